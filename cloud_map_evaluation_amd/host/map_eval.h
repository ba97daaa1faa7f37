// map_eval.h — host side of the drop-in: the reference's Param / MapEval surface (map_eval/src/map_eval.h:60-362) with the
// metric hot path delegated to libmapeval_hip.so through the C ABI (include/mapeval_hip.h).  No Open3D / Eigen / PCL /
// TBB / yaml-cpp: clouds are std::vector<double> (AoS xyz, the same memory layout as open3d PointCloud::points_).
#pragma once

#include <array>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mapeval_hip.h"
#include "dist_comm.hpp"  // medist::Comm, medist::DevMem (multi-GPU: num_gpus > 1)

using Vector5d = std::array<double, 5>;

struct PointCloud {  // stands in for open3d::geometry::PointCloud on the hot path (points_ only)
    std::vector<double> points_;  // xyz xyz ...
    size_t size() const { return points_.size() / 3; }
    bool IsEmpty() const { return points_.empty(); }
};

// Same members and defaults as the reference's Param (map_eval.h:60-116); YAML keys in map_eval_main.cpp:120-208.
struct Param {
    std::string evaluation_map_pcd_path_ = "/data/map_evaluation/canteen/";
    std::string map_gt_path_ = "/data/map_evaluation/canteen/merged_scan.pcd";
    std::string result_path_ = "/home/hts/workspace/dataset/eva_results/";
    std::string pcd_file_name_ = "map.pcd";
    std::string name_;
    int evaluation_method_ = 2;
    double voxel_size_ = 1.0;
    double icp_max_distance_ = 2.5;
    double nn_radius_ = 0.2;
    bool save_immediate_result_ = false;
    bool evaluate_mme_ = true;
    bool evaluate_gt_mme_ = true;
    bool evaluate_using_initial_ = true;
    bool evaluate_noised_gt_ = false;
    bool use_visualization = false;
    bool enable_debug = false;
    bool use_tbb_mme = true;  // accepted; the GPU path has a single implementation
    Vector5d trunc_dist_{{0.2, 0.1, 0.08, 0.05, 0.01}};  // the reference leaves this uninitialised (map_eval.h:85)
    std::array<double, 16> initial_matrix_{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};  // row-major 4x4
    double noise_std_dev_ = 0.1;
    double vmd_voxel_size_ = 3.0;
    double downsample_size = 0.01;
    // ---- new, optional keys (absent in the reference's configs) ----
    int gpu_device = 0;             // `gpu_device:` HIP device ordinal
    bool strict_reference = false;  // `strict_reference:` true = reproduce FULL CD = 0 on the initial-matrix path
                                    // (the reference never calls computeChamferDistance there, map_eval.cpp:1204-1260)
    int num_gpus = 1;               // `num_gpus:` N > 1 = one process per GPU (devices gpu_device .. gpu_device + N - 1), the
                                    // clouds cut into N slabs, collectives over RCCL (map_eval_dist.cpp)
    int dist_rank = 0;              // (set by the launcher, not a YAML key)
    void printParam() const;
};

Param loadParametersFromYAML(const std::string &yaml_file_path);  // map_eval_main.cpp:120-208
std::string paramToJson(const Param &p);                          // for tests (--parse-config)

class MapEval {
public:
    explicit MapEval(Param &param);  // map_eval.h:123-189: results folder + map_results.txt header (append mode)
    ~MapEval();

    int process();                                         // map_eval.cpp:4-102
    int processOneCall(bool from_host, double t_loaded);   // its metric phase (:52-85) as one me_run_suite_from call
    void computeMME(PointCloud &cloud, PointCloud &gt);    // map_eval.cpp:149-189
    void calculateMetricsWithInitialMatrix();              // map_eval.cpp:1204-1260
    void finishInitialMatrixMetrics(const me_nn_stats_out &eg, const me_nn_stats_out &ge, double t_acc_s);  // its tail (:1238-1259)
    // map_eval.cpp:191-237, registration_methods 0 / 1 / 2; with a communicator: queries sharded, sums all-reduced (map_eval_dist.cpp);
    // metrics = false: stop after the loop (the distributed host computes the ICP-path statistics on its slabs)
    int performRegistration(bool metrics = true);
    void finishRegistrationMetrics(const me_nn_stats_out &eg, const me_nn_stats_out &ge, double t_acc_s);  // calculateMetrics' tail
    void calculateMetrics();                               // map_eval.cpp:1147-1202
    double computeChamferDistance();                       // map_eval.cpp:1398-1431
    void calculateVMD(bool tables_ready = false, bool write_files = true);  // map_eval.cpp:240-390
    bool renderEntropy(int slot, std::vector<double> &xyz, std::vector<double> &rgb, bool want_points);  // :686-735
    void saveMmeResults();                                 // map_eval.cpp:392-421
    void saveRegistrationResults();                        // map_eval.cpp:424-482 (text lines; renderers out of scope)

    // multi-GPU (map_eval_dist.cpp): the communicator of this rank; forced = take the distributed path with one rank too
    void setComm(medist::Comm *comm, bool forced) {
        comm_ = comm;
        dist_forced_ = forced;
    }
    int processDist(double t_loaded);

    Param param_;
    // results, same names as the reference (map_eval.h:328-353)
    std::vector<Vector5d> est_gt_results, gt_est_results;
    Vector5d f1_vec{{0, 0, 0, 0, 0}}, cd_vec{{0, 0, 0, 0, 0}}, iou_vec{{0, 0, 0, 0, 0}};
    std::array<double, 16> trans{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};  // ICP result (row-major), map_eval.h:332
    double vmd = 0.0, full_chamfer_dist = 0.0, scs_overall = 0.0;
    double mme_est = 0.0, mme_gt = 0.0, max_abs_entropy = 0.0, min_abs_entropy = 0.0;
    std::vector<double> est_entropies, gt_entropies;
    std::vector<uint8_t> valid_entropy_points;
    std::vector<double> map_entropy_xyz, map_entropy_rgb, gt_entropy_xyz, gt_entropy_rgb;  // map_3d_entropy / gt_3d_entropy (:330)
    std::string last_error;

private:
    int fail(const std::string &msg);
    int allReduceHost(std::vector<double> &v, bool min_op);
    int gatherPerPoint(int slot, size_t n_global, const std::vector<int64_t> &tags, const std::vector<double> *vals,
                       const std::vector<uint8_t> *flags, std::vector<double> *vals_out, std::vector<uint8_t> *flags_out);
    int exchangeCloud(const std::vector<double> &pts, size_t i0, size_t i1, const double *T, int axis, const std::vector<double> &cuts,
                      double halo, medist::DevMem &recv, std::vector<int64_t> &tags_out, int64_t *n_recv);
    int reduceIcp(me_icp_sums &s, me_icp_lsq &q, int method);  // all-reduce of the registration step's additive sums
    medist::DevMem pool_[7];  // device buffers of the collectives, kept for the run
    medist::Comm *comm_ = nullptr;
    bool dist_forced_ = false;
    me_ctx *render_ctx_ = nullptr;  // rank 0 of a multi-GPU run: the whole clouds, for the colour renderers
    std::shared_ptr<PointCloud> map_3d_, gt_3d_;
    me_ctx *ctx_ = nullptr;
    double t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t_fcd = 0, t_acc = 0;
    double t_vmd = 0, t_v = 0, t_cdf = 0, t_scs = 0;
    std::string subfolder, results_subfolder, results_file_path;
    std::ofstream file_result;
};
