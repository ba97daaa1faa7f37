// ref_driver.cpp — C face of oracle/_ref/libmapeval_ref.so.  TEST INFRASTRUCTURE ONLY.
//
// libmapeval_ref.so = the reference's OWN map_eval.cpp + voxel_calculator.cpp, compiled unmodified from /root/reference
// (see Makefile; nothing of them is copied into this repository) against the functional stand-in headers under standin/
// (Eigen, Open3D, TBB, PCL and yaml-cpp are absent from this image), plus this file, which only
//   * builds a `Param` and a `MapEval`, hands it the two clouds, calls the reference's member functions in the order
//     `MapEval::process()` calls them (map_eval.cpp:52-85) — or `process()` itself on PCD files — and copies the members
//     that hold the results out through plain C pointers;
//   * exposes single reference functions (getDiffRegResult*, computeChamferDistance, ComputeMeanMapEntropy*,
//     VoxelCalculator::buildVoxelMap / computeWassersteinDistanceGaussian, calculateVMD) for per-row tests.
// No metric arithmetic lives here.  What is the builder's and not the reference's: the stand-in headers.
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

// The clouds and the result members of MapEval are private (map_eval.h:321-353) and VoxelCalculator::getVoxelIndex is too;
// the driver has to reach them.  Access specifiers do not change the layout under the Itanium ABI for these classes.
#define private public
#include "map_eval.h"
#undef private

extern "C" {

typedef struct ref_config {
    double trunc[5];          // accuracy_level
    double icp_max_distance;  // icp_max_distance
    double nn_radius;         // nn_radius
    double vmd_voxel_size;    // vmd_voxel_size
    double downsample_size;   // downsample_size (process() only)
    double T[16];             // initial_matrix, row-major
    int32_t evaluate_mme, evaluate_gt_mme, use_tbb_mme, evaluate_using_initial, save_immediate_result, registration_methods;
} ref_config;

typedef struct ref_results {
    int64_t n_est, n_gt;      // points_.size() after process()'s down-sampling / as handed over
    int64_t n_est_gt_vecs, n_gt_est_vecs;  // how many Vector5d the reference pushed (5 or 4)
    double est_gt[5][5];      // est_gt_results: [mean, rmse, fitness, sigma, number][threshold]
    double gt_est[5][5];      // gt_est_results (as the reference computes them: its own pairing, finding 4 of SURVEY)
    double cd_vec[5], f1_vec[5], iou_vec[5];
    double mme_est, mme_gt, min_abs_entropy, max_abs_entropy;
    double vmd, scs, full_chamfer_dist;
    double trans[16];
} ref_results;

}  // extern "C"

namespace {

Param make_param(const ref_config &c, const std::string &workdir, const std::string &gt_path = "") {
    Param p;
    p.evaluation_method_ = c.registration_methods;
    p.icp_max_distance_ = c.icp_max_distance;
    for (int i = 0; i < 5; ++i) p.trunc_dist_[i] = c.trunc[i];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) p.initial_matrix_(i, j) = c.T[4 * i + j];
    p.save_immediate_result_ = c.save_immediate_result != 0;
    p.evaluate_mme_ = c.evaluate_mme != 0;
    p.evaluate_gt_mme_ = c.evaluate_gt_mme != 0;
    p.evaluate_using_initial_ = c.evaluate_using_initial != 0;
    p.use_tbb_mme = c.use_tbb_mme != 0;
    p.nn_radius_ = c.nn_radius;
    p.vmd_voxel_size_ = c.vmd_voxel_size;
    p.downsample_size = c.downsample_size;
    p.evaluation_map_pcd_path_ = workdir;
    if (!p.evaluation_map_pcd_path_.empty() && p.evaluation_map_pcd_path_.back() != '/') p.evaluation_map_pcd_path_ += '/';
    p.map_gt_path_ = gt_path;
    p.name_ = "oracle_ref";
    p.result_path_ = p.evaluation_map_pcd_path_ + "map_results/";
    p.pcd_file_name_ = "global_pcd_lidar.pcd";  // one of the names MapEval's constructor accepts without complaint
    p.use_visualization = false;
    p.enable_debug = false;
    return p;
}

void set_cloud(PointCloud &pc, const double *xyz, int64_t n) {
    pc.points_.resize((size_t) n);
    for (int64_t i = 0; i < n; ++i) pc.points_[(size_t) i] = Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

void copy_vecs(const std::vector<Vector5d> &v, double out[5][5], int64_t *count) {
    *count = (int64_t) v.size();
    for (int r = 0; r < 5; ++r)
        for (int k = 0; k < 5; ++k) out[r][k] = (size_t) r < v.size() ? v[(size_t) r][k] : 0.0;
}

void collect(MapEval &me, ref_results *out) {
    out->n_est = (int64_t) me.map_3d_->points_.size();
    out->n_gt = (int64_t) me.gt_3d_->points_.size();
    copy_vecs(me.est_gt_results, out->est_gt, &out->n_est_gt_vecs);
    copy_vecs(me.gt_est_results, out->gt_est, &out->n_gt_est_vecs);
    for (int k = 0; k < 5; ++k) {
        out->cd_vec[k] = me.cd_vec[k];
        out->f1_vec[k] = me.f1_vec[k];
        out->iou_vec[k] = me.iou_vec[k];
    }
    out->mme_est = me.mme_est;
    out->mme_gt = me.mme_gt;
    out->min_abs_entropy = me.min_abs_entropy;
    out->max_abs_entropy = me.max_abs_entropy;
    out->vmd = me.vmd;
    out->scs = me.scs_overall;
    out->full_chamfer_dist = me.full_chamfer_dist;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out->trans[4 * i + j] = me.trans(i, j);
}

void copy_doubles(const std::vector<double> &v, double *out) {
    if (out && !v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(double));
}

thread_local std::string g_err;

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    } catch (...) {
        g_err = "unknown exception";
        return -1;
    }
}

}  // namespace

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

// The body of MapEval::process() after loading and down-sampling (map_eval.cpp:51-85), on clouds handed over in memory:
// computeMME -> calculateMetricsWithInitialMatrix -> calculateVMD (-> saveMmeResults / saveRegistrationResults when asked).
// est is transformed in place by the reference (:1206); est_out (nullable, n_est x 3) receives it.
int ref_suite_initial(const double *est, int64_t n_est, const double *gt, int64_t n_gt, const ref_config *cfg,
                      const char *workdir, ref_results *out, double *est_entropies, double *gt_entropies, double *est_out) {
    return guarded([&] {
        Param p = make_param(*cfg, workdir);
        MapEval me(p);
        set_cloud(*me.map_3d_, est, n_est);
        set_cloud(*me.gt_3d_, gt, n_gt);
        if (me.param_.evaluate_mme_) {
            me.computeMME(me.map_3d_, me.gt_3d_);
            if (me.param_.save_immediate_result_) me.saveMmeResults();
        }
        me.calculateMetricsWithInitialMatrix();
        me.calculateVMD();
        if (me.param_.save_immediate_result_) me.saveRegistrationResults();
        collect(me, out);
        copy_doubles(me.est_entropies, est_entropies);
        copy_doubles(me.gt_entropies, gt_entropies);
        if (est_out)
            for (size_t i = 0; i < me.map_3d_->points_.size(); ++i)
                for (int d = 0; d < 3; ++d) est_out[3 * i + d] = me.map_3d_->points_[i](d);
    });
}

// MapEval::process() itself (map_eval.cpp:4-104) on files: <workdir>/global_pcd_lidar.pcd and gt_path.
int ref_process(const ref_config *cfg, const char *workdir, const char *gt_path, ref_results *out, int *process_rc) {
    return guarded([&] {
        Param p = make_param(*cfg, workdir, gt_path);
        MapEval me(p);
        *process_rc = me.process();
        collect(me, out);
    });
}

// The ICP-path statistics (calculateMetrics, map_eval.cpp:1147-1202) on an already registered pair: the correspondence set a
// converged RegistrationICP would return for identity motion (EvaluateRegistration(map, gt, icp_max_distance)).
int ref_calculate_metrics(const double *est, int64_t n_est, const double *gt, int64_t n_gt, const ref_config *cfg,
                          const char *workdir, ref_results *out, int64_t *n_corr) {
    return guarded([&] {
        Param p = make_param(*cfg, workdir);
        MapEval me(p);
        set_cloud(*me.map_3d_, est, n_est);
        set_cloud(*me.gt_3d_, gt, n_gt);
        auto reg = pipelines::registration::EvaluateRegistration(*me.map_3d_, *me.gt_3d_, me.param_.icp_max_distance_);
        *n_corr = (int64_t) reg.correspondence_set_.size();
        me.calculateMetrics(reg);
        collect(me, out);
    });
}

// getDiffRegResult variants on a caller-supplied correspondence set: variant 0 = getDiffRegResultWithCorrespondence
// (map_eval.cpp:1069-1145), 1 = 6-argument getDiffRegResult (:990-1067), 2 = 4-argument getDiffRegResult (:828-897).
int ref_diff_reg_result(int variant, const double *src, int64_t n_src, const double *tgt, int64_t n_tgt, const int32_t *pairs,
                        int64_t n_pairs, const double trunc[5], const char *workdir, double out[5][5], int64_t *n_vecs) {
    return guarded([&] {
        ref_config c{};
        for (int k = 0; k < 5; ++k) c.trunc[k] = trunc[k];
        for (int i = 0; i < 4; ++i) c.T[5 * i] = 1.0;
        Param p = make_param(c, workdir);
        MapEval me(p);
        PointCloud s, t, sset, tset;
        set_cloud(s, src, n_src);
        set_cloud(t, tgt, n_tgt);
        pipelines::registration::CorrespondenceSet cs((size_t) n_pairs);
        for (int64_t i = 0; i < n_pairs; ++i) cs[(size_t) i] = Eigen::Vector2i(pairs[2 * i], pairs[2 * i + 1]);
        std::vector<Vector5d> res;
        if (variant == 0)
            me.getDiffRegResultWithCorrespondence(res, cs, s, t, sset, tset);
        else if (variant == 1)
            me.getDiffRegResult(res, cs, s, t, sset, tset);
        else
            me.getDiffRegResult(res, cs, s, t);
        copy_vecs(res, out, n_vecs);
    });
}

// computeChamferDistance (map_eval.cpp:1398-1431)
int ref_chamfer(const double *a, int64_t na, const double *b, int64_t nb, const char *workdir, double *cd) {
    return guarded([&] {
        ref_config c{};
        Param p = make_param(c, workdir);
        MapEval me(p);
        PointCloud ca, cb;
        set_cloud(ca, a, na);
        set_cloud(cb, b, nb);
        *cd = me.computeChamferDistance(ca, cb);
    });
}

// MME loops on one cloud with a fresh MapEval (so that valid_entropy_points is this call's alone):
// variant 0 = ComputeMeanMapEntropy (serial, k >= 5, :1438-1535), 1 = ...UsingNormal (OpenMP, k >= 10, :1538-1606),
// 2 = ...UsingNormalTBB (k >= 10, :1608-1737).  entropies[n], valid[n] nullable.
// valid_raw (nullable): the bits of valid_entropy_points exactly as the reference's loop left them, race included.
int ref_mme_raw(int variant, const double *xyz, int64_t n, double radius, const char *workdir, double *entropies, uint8_t *valid,
                uint8_t *valid_raw, double *mean);
int ref_mme(int variant, const double *xyz, int64_t n, double radius, const char *workdir, double *entropies, uint8_t *valid,
            double *mean) {
    return ref_mme_raw(variant, xyz, n, radius, workdir, entropies, valid, nullptr, mean);
}
int ref_mme_raw(int variant, const double *xyz, int64_t n, double radius, const char *workdir, double *entropies, uint8_t *valid,
                uint8_t *valid_raw, double *mean) {
    return guarded([&] {
        ref_config c{};
        Param p = make_param(c, workdir);
        MapEval me(p);
        auto pc = std::make_shared<PointCloud>();
        set_cloud(*pc, xyz, n);
        std::vector<double> ent;
        if (variant == 0)
            *mean = me.ComputeMeanMapEntropy(pc, ent, radius);
        else if (variant == 1)
            *mean = me.ComputeMeanMapEntropyUsingNormal(pc, ent, radius);
        else
            *mean = me.ComputeMeanMapEntropyUsingNormalTBB(pc, ent, radius);
        copy_doubles(ent, entropies);
        // The parallel variants set bits of `valid_entropy_points` (a std::vector<bool>) from several threads: a data race on
        // shared words that can lose an update (map_eval.cpp:1586, :1694).  A point is valid exactly when its entropy was
        // stored (the same branch, :1692-1697; a stored entropy of exactly 0.0 would need det = 1 / (2 pi e) to the last bit),
        // so the flag reported here is what a race-free run of the reference leaves.
        if (valid)
            for (int64_t i = 0; i < n; ++i) valid[i] = (me.valid_entropy_points[(size_t) i] || ent[(size_t) i] != 0.0) ? 1 : 0;
        if (valid_raw)
            for (int64_t i = 0; i < n; ++i) valid_raw[i] = me.valid_entropy_points[(size_t) i] ? 1 : 0;
    });
}

// ComputeEntropy (map_eval.cpp:1433-1436); cov row-major
double ref_compute_entropy(const double cov[9]) {
    ref_config c{};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static std::unique_ptr<MapEval> me;
    if (!me) {
        Param p = make_param(c, std::filesystem::temp_directory_path().string() + "/");
        me.reset(new MapEval(p));
    }
    Eigen::Matrix3d m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m(i, j) = cov[3 * i + j];
    return me->ComputeEntropy(m);
}

// calculateVMD (map_eval.cpp:240-390) alone; voxel_errors.txt and voxel_wasserstein_cdf.txt land in <workdir>/map_results/.
int ref_vmd(const double *est, int64_t n_est, const double *gt, int64_t n_gt, double voxel, const char *workdir, double *vmd,
            double *scs) {
    return guarded([&] {
        ref_config c{};
        c.vmd_voxel_size = voxel;
        Param p = make_param(c, workdir);
        MapEval me(p);
        set_cloud(*me.map_3d_, est, n_est);
        set_cloud(*me.gt_3d_, gt, n_gt);
        me.calculateVMD();
        *vmd = me.vmd;
        *scs = me.scs_overall;
    });
}

// ---- VoxelCalculator (voxel_calculator.cpp) -------------------------------------------------------------------
struct ref_voxelmap {
    VoxelCalculator calc;
    std::vector<std::pair<Eigen::Vector3i, const VoxelInfo *>> sorted;  // ascending (ix, iy, iz)
    explicit ref_voxelmap(double v) : calc(v) {}
};

ref_voxelmap *ref_voxel_build(const double *xyz, int64_t n, double voxel) {
    ref_voxelmap *h = nullptr;
    const int rc = guarded([&] {
        h = new ref_voxelmap(voxel);
        PointCloud pc;
        set_cloud(pc, xyz, n);
        h->calc.buildVoxelMap(pc);  // voxel_calculator.cpp:21-56 (+ computeVoxelEntropy :97-113, getVoxelIndex :241-245)
        for (const auto &kv : h->calc.getVoxelMap()) h->sorted.emplace_back(kv.first, &kv.second);
        std::sort(h->sorted.begin(), h->sorted.end(), [](const auto &a, const auto &b) {
            for (int d = 0; d < 3; ++d)
                if (a.first[d] != b.first[d]) return a.first[d] < b.first[d];
            return false;
        });
    });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}
void ref_voxel_free(ref_voxelmap *h) { delete h; }
int64_t ref_voxel_count(const ref_voxelmap *h) { return (int64_t) h->sorted.size(); }
// keys[V][3], npts[V], mu[V][3], sigma[V][9] row-major AS STORED, entropy[V], energy[V], active[V]; any may be NULL
void ref_voxel_export(const ref_voxelmap *h, int32_t *keys, int32_t *npts, double *mu, double *sigma, double *entropy,
                      double *energy, int32_t *active) {
    for (size_t v = 0; v < h->sorted.size(); ++v) {
        const VoxelInfo &vi = *h->sorted[v].second;
        for (int d = 0; d < 3; ++d) {
            if (keys) keys[3 * v + d] = h->sorted[v].first[d];
            if (mu) mu[3 * v + d] = vi.mu(d);
        }
        if (npts) npts[v] = vi.num_points;
        if (sigma)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) sigma[9 * v + 3 * i + j] = vi.sigma(i, j);
        if (entropy) entropy[v] = vi.entropy;
        if (energy) energy[v] = vi.energy;
        if (active) active[v] = vi.active;
    }
}
// est.updateVoxelMap(gt.getVoxelMap()) (voxel_calculator.cpp:142-172) -> counts[3] = active / old / new, recounted
// from the labels the reference left in the estimated map (it only prints its own counters).
void ref_voxel_update(ref_voxelmap *est, const ref_voxelmap *gt, int64_t counts[3]) {
    est->calc.updateVoxelMap(gt->calc.getVoxelMap());
    counts[0] = counts[1] = counts[2] = 0;
    for (const auto &kv : est->calc.getVoxelMap()) {
        if (kv.second.active == 1) counts[0]++;
        if (kv.second.active == 0) counts[1]++;
        if (kv.second.active == 2) counts[2]++;
    }
}
// computeWassersteinDistanceGaussian(voxel1, voxel2) (voxel_calculator.cpp:115-140); sigma = stored 3x3, row-major
double ref_w2_gaussian(const double mu1[3], const double sigma1[9], int n1, const double mu2[3], const double sigma2[9], int n2) {
    VoxelCalculator calc(1.0);
    VoxelInfo a, b;
    for (int i = 0; i < 3; ++i) {
        a.mu(i) = mu1[i];
        b.mu(i) = mu2[i];
        for (int j = 0; j < 3; ++j) {
            a.sigma(i, j) = sigma1[3 * i + j];
            b.sigma(i, j) = sigma2[3 * i + j];
        }
    }
    a.num_points = n1;
    b.num_points = n2;
    return calc.computeWassersteinDistanceGaussian(a, b);
}
// getVoxelIndex (voxel_calculator.cpp:241-245)
void ref_voxel_index(const double p[3], double voxel, int32_t out[3]) {
    VoxelCalculator calc(voxel);
    const Eigen::Vector3i k = calc.getVoxelIndex(Eigen::Vector3d(p[0], p[1], p[2]));
    for (int d = 0; d < 3; ++d) out[d] = k[d];
}
// getNeighborIndices (voxel_calculator.cpp:7-19) -> number of neighbours; out[count][3] nullable
int64_t ref_neighbor_indices(const int32_t index[3], int radius, int32_t *out) {
    VoxelCalculator calc(1.0);
    const auto v = calc.getNeighborIndices(Eigen::Vector3i(index[0], index[1], index[2]), radius);
    if (out)
        for (size_t i = 0; i < v.size(); ++i)
            for (int d = 0; d < 3; ++d) out[3 * i + d] = v[i][d];
    return (int64_t) v.size();
}

// ---- the stand-in KD-tree on its own (builder's code; cross-checked against the oracle's tree and brute force) ----
void ref_kdtree_nn1(const double *ref, int64_t n, const double *q, int64_t m, int32_t *idx, double *d2) {
    PointCloud pc;
    set_cloud(pc, ref, n);
    geometry::KDTreeFlann tree(pc);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < m; ++i) {
        std::vector<int> ii(1);
        std::vector<double> dd(1);
        tree.SearchKNN(Eigen::Vector3d(q[3 * i], q[3 * i + 1], q[3 * i + 2]), 1, ii, dd);
        if (idx) idx[i] = ii[0];
        if (d2) d2[i] = dd[0];
    }
}
void ref_kdtree_radius_count(const double *ref, int64_t n, const double *q, int64_t m, double r, int32_t *count) {
    PointCloud pc;
    set_cloud(pc, ref, n);
    geometry::KDTreeFlann tree(pc);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < m; ++i) {
        std::vector<int> ii;
        std::vector<double> dd;
        count[i] = tree.SearchRadius(Eigen::Vector3d(q[3 * i], q[3 * i + 1], q[3 * i + 2]), r, ii, dd);
    }
}

}  // extern "C"
