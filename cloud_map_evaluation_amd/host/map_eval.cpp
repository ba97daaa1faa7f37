// map_eval.cpp — host orchestration mirroring MapEval::process (map_eval/src/map_eval.cpp:4-102): same config keys,
// same result-file lines, same output file names; every metric comes from libmapeval_hip.so (no CPU metric path).
#include "map_eval.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <ctime>
#include <filesystem>
#include <iomanip>
#include <iostream>
#include <numeric>
#include <sstream>

#include "pcd_io.hpp"
#include "yaml_lite.hpp"

namespace fs = std::filesystem;

namespace {

struct TicToc {  // include/tic_toc.h:10-24 — milliseconds since construction
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double toc() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// Eigen's default IOFormat for `os << v.transpose()`: entries right-aligned to the widest one, separated by one space.
std::string eigen_row(const Vector5d &v, int precision) {
    std::vector<std::string> s;
    size_t w = 0;
    for (double x : v) {
        std::ostringstream o;
        o << std::fixed << std::setprecision(precision) << x;
        s.push_back(o.str());
        w = std::max(w, s.back().size());
    }
    std::string out;
    for (size_t i = 0; i < s.size(); ++i) {
        if (i) out += " ";
        out += std::string(w - s[i].size(), ' ') + s[i];
    }
    return out;
}

// Eigen's default IOFormat for `os << Matrix4d` (row-major input): common width over all 16 coefficients
std::string eigen_matrix4(const double *m, int precision) {
    std::string s[16];
    size_t w = 0;
    for (int i = 0; i < 16; ++i) {
        std::ostringstream o;
        o << std::fixed << std::setprecision(precision) << m[i];
        s[i] = o.str();
        w = std::max(w, s[i].size());
    }
    std::string out;
    for (int r = 0; r < 4; ++r) {
        for (int c = 0; c < 4; ++c) out += (c ? " " : "") + std::string(w - s[4 * r + c].size(), ' ') + s[4 * r + c];
        if (r < 3) out += "\n";
    }
    return out;
}

Vector5d to5(const double *p) { return Vector5d{{p[0], p[1], p[2], p[3], p[4]}}; }

void push_results(std::vector<Vector5d> &dst, const me_nn_stats_out &o) {
    // result.push_back(mean, rmse, fitness, sigma, number) (map_eval.cpp:1140-1144)
    dst.clear();
    dst.push_back(to5(o.mean));
    dst.push_back(to5(o.rmse));
    dst.push_back(to5(o.fitness));
    dst.push_back(to5(o.sigma));
    dst.push_back(to5(o.number));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
void Param::printParam() const {
    std::cout << "\n[==================== Experiment Configuration ====================]\n"
              << "Experiment Name: " << name_ << "\n"
              << "Evaluation Map Path: " << evaluation_map_pcd_path_ << "\n"
              << "Ground Truth Map Path: " << map_gt_path_ << "\n"
              << "Result Save Path: " << result_path_ << "\n"
              << "ICP Maximum Distance: " << icp_max_distance_ << "\n"
              << "Evaluation Method: " << evaluation_method_ << "\n"
              << "Truncation Distance: " << eigen_row(trunc_dist_, 6) << "\n"
              << "Save Immediate Result: " << save_immediate_result_ << "\n"
              << "Evaluate MME: " << evaluate_mme_ << "\n"
              << "Evaluate Ground Truth MME: " << evaluate_gt_mme_ << "\n"
              << "Nearest Neighbor Radius: " << nn_radius_ << "\n"
              << "Use Initial Matrix for Evaluation: " << evaluate_using_initial_ << "\n"
              << "Voxel Size for VMD: " << vmd_voxel_size_ << "\n"
              << "GPU device: " << gpu_device << "\n"
              << "[====================================================================]\n";
}

Param loadParametersFromYAML(const std::string &yaml_file_path) {
    // Same required / optional key split as map_eval_main.cpp:120-208 (a missing required key throws).
    const yaml_lite::Document config = yaml_lite::Document::load_file(yaml_file_path);
    Param param;
    param.evaluation_method_ = config.as_int("registration_methods");
    param.icp_max_distance_ = config.as_double("icp_max_distance");
    if (config.has("accuracy_level") && config.at("accuracy_level").seq.size() >= 5)
        for (int i = 0; i < 5; ++i) param.trunc_dist_[i] = yaml_lite::Document::to_double(config.at("accuracy_level").seq[i], "accuracy_level");
    if (config.has("initial_matrix") && config.at("initial_matrix").rows.size() >= 4)
        for (int i = 0; i < 4; ++i) {
            const auto &row = config.at("initial_matrix").rows[i];
            if (row.size() < 4) throw std::runtime_error("initial_matrix: each row needs 4 numbers");
            for (int j = 0; j < 4; ++j) param.initial_matrix_[4 * i + j] = yaml_lite::Document::to_double(row[j], "initial_matrix");
        }
    param.save_immediate_result_ = config.as_bool("save_immediate_result");
    param.evaluate_mme_ = config.as_bool("evaluate_mme");
    param.evaluate_gt_mme_ = config.as_bool("evaluate_gt_mme");
    param.evaluate_using_initial_ = config.as_bool("evaluate_using_initial");
    param.nn_radius_ = config.as_double("nn_radius");
    param.vmd_voxel_size_ = config.as_double("vmd_voxel_size");
    param.downsample_size = config.as_double("downsample_size");
    param.evaluation_map_pcd_path_ = config.as_string("estimate_map_path");
    param.map_gt_path_ = config.as_string("gt_map_path");
    param.name_ = config.as_string("scene_name");
    if (!param.evaluation_map_pcd_path_.empty() && param.evaluation_map_pcd_path_.back() != '/') param.evaluation_map_pcd_path_ += '/';
    param.result_path_ = param.evaluation_map_pcd_path_ + "map_results/";
    if (config.has("pcd_file_name")) param.pcd_file_name_ = config.as_string("pcd_file_name");
    if (config.has("evaluate_noised_gt")) param.evaluate_noised_gt_ = config.as_bool("evaluate_noised_gt");
    if (config.has("noise_std_dev")) param.noise_std_dev_ = config.as_double("noise_std_dev");
    if (config.has("voxel_size")) param.voxel_size_ = config.as_double("voxel_size");
    if (config.has("use_visualization")) param.use_visualization = config.as_bool("use_visualization");
    param.enable_debug = config.as_bool("enable_debug");
    if (config.has("use_tbb_mme")) param.use_tbb_mme = config.as_bool("use_tbb_mme");
    if (config.has("gpu_device")) param.gpu_device = config.as_int("gpu_device");
    if (config.has("strict_reference")) param.strict_reference = config.as_bool("strict_reference");
    if (config.has("num_gpus")) param.num_gpus = config.as_int("num_gpus");
    if (param.num_gpus < 1 || param.num_gpus > 64) throw std::runtime_error("num_gpus must be in 1..64");
    return param;
}

std::string paramToJson(const Param &p) {
    std::ostringstream o;
    o << std::setprecision(17);
    auto b = [](bool v) { return v ? "true" : "false"; };
    o << "{\"registration_methods\": " << p.evaluation_method_ << ", \"icp_max_distance\": " << p.icp_max_distance_
      << ", \"accuracy_level\": [";
    for (int i = 0; i < 5; ++i) o << (i ? ", " : "") << p.trunc_dist_[i];
    o << "], \"initial_matrix\": [";
    for (int i = 0; i < 16; ++i) o << (i ? ", " : "") << p.initial_matrix_[i];
    o << "], \"save_immediate_result\": " << b(p.save_immediate_result_) << ", \"evaluate_mme\": " << b(p.evaluate_mme_)
      << ", \"evaluate_gt_mme\": " << b(p.evaluate_gt_mme_) << ", \"evaluate_using_initial\": " << b(p.evaluate_using_initial_)
      << ", \"nn_radius\": " << p.nn_radius_ << ", \"vmd_voxel_size\": " << p.vmd_voxel_size_
      << ", \"downsample_size\": " << p.downsample_size << ", \"estimate_map_path\": \"" << p.evaluation_map_pcd_path_
      << "\", \"gt_map_path\": \"" << p.map_gt_path_ << "\", \"scene_name\": \"" << p.name_ << "\", \"pcd_file_name\": \""
      << p.pcd_file_name_ << "\", \"enable_debug\": " << b(p.enable_debug) << ", \"use_tbb_mme\": " << b(p.use_tbb_mme)
      << ", \"use_visualization\": " << b(p.use_visualization) << ", \"result_path\": \"" << p.result_path_
      << "\", \"gpu_device\": " << p.gpu_device << ", \"strict_reference\": " << b(p.strict_reference) << ", \"num_gpus\": " << p.num_gpus
      << "}";
    return o.str();
}

// ---------------------------------------------------------------------------------------------------------------
MapEval::MapEval(Param &param) : param_(param), map_3d_(new PointCloud), gt_3d_(new PointCloud) {
    // results sub-folder by file name (map_eval.h:143-155), created next to the estimated map (:158-164)
    if (param_.pcd_file_name_ == "merged_maps_all_trans.pcd") subfolder = "merged_maps_all_results/";
    else if (param_.pcd_file_name_ == "merged_maps_s0_trans.pcd") subfolder = "merged_maps_s0_results/";
    else if (param_.pcd_file_name_ == "merged_maps_s1_trans.pcd") subfolder = "merged_maps_s1_results/";
    else subfolder = "map_results/";
    results_subfolder = param_.evaluation_map_pcd_path_ + subfolder;
    std::cout << "INFO: Saving results to: " << results_subfolder << std::endl;
    std::error_code ec;
    if (!fs::exists(results_subfolder)) fs::create_directory(results_subfolder, ec);
    results_file_path = results_subfolder + "map_results.txt";
    if (param_.dist_rank > 0) results_file_path = "/dev/null";  // multi-GPU: rank 0 alone writes the result files
    file_result.open(results_file_path, std::ios::app);  // append mode (:168)
    if (!file_result.is_open()) std::cerr << "ERROR: Failed to open results file at " << results_file_path << std::endl;
    const std::time_t now_c = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
    std::stringstream time_stream;
    time_stream << std::put_time(std::localtime(&now_c), "%Y-%m-%d %X");
    file_result << param_.name_ << " ===================== " << time_stream.str() << " ===================== " << std::endl;
    file_result << "Ground Truth Path: " << param_.map_gt_path_ << std::endl;
    file_result << "Evaluation Map Path: " << param_.evaluation_map_pcd_path_ + param_.pcd_file_name_ << std::endl;
    std::cout << "INFO: Evaluation details saved to " << results_file_path << std::endl;
}

MapEval::~MapEval() {
    if (render_ctx_) me_destroy(render_ctx_);
    if (ctx_) me_destroy(ctx_);
    file_result.close();
}

int MapEval::fail(const std::string &msg) {
    last_error = msg;
    std::cerr << "ERROR: " << msg << std::endl;
    return -1;
}

int MapEval::process() {
    TicToc tic_toc;
    std::string err;
    // ground truth: .pcd or .ply by extension (map_eval.cpp:9-17)
    const std::string ext = param_.map_gt_path_.substr(param_.map_gt_path_.find_last_of(".") + 1);
    bool ok_gt;
    std::vector<double> gt_normals;  // normal_x/y/z of the ground truth, if the file has them (point-to-plane ICP needs them)
    if (ext == "pcd") ok_gt = pcio::read_pcd(param_.map_gt_path_, gt_3d_->points_, &err, &gt_normals);
    else if (ext == "ply") ok_gt = pcio::read_ply(param_.map_gt_path_, gt_3d_->points_, &err, &gt_normals);
    else return fail("Unsupported ground truth file format: " + param_.map_gt_path_);
    if (!ok_gt) std::cerr << "WARNING: " << err << std::endl;
    // the estimated map's own normal_x/y/z, if its PCD has them: Open3D's InitializePointCloudForGeneralizedICP uses a
    // cloud's normals when it carries them and estimates them (KNN 20) only otherwise
    std::vector<double> map_normals;
    const bool success = pcio::read_pcd(param_.evaluation_map_pcd_path_ + param_.pcd_file_name_, map_3d_->points_, &err, &map_normals);
    if (param_.enable_debug)
        std::cout << "INFO: Loading map point cloud from: " << param_.evaluation_map_pcd_path_ + param_.pcd_file_name_ << std::endl;
    if (!success) return fail("Failed to load point cloud from the specified path.");
    if (map_3d_->IsEmpty() || gt_3d_->IsEmpty()) return fail("One or both point clouds are empty!");

    // ---- the GPU engine: no CPU fallback ----
    ctx_ = me_create(param_.gpu_device, 0);
    if (!ctx_) {
        file_result << std::fixed << std::setprecision(15) << "Estimated-Ground Truth point count: " << map_3d_->size() << " / "
                    << gt_3d_->size() << std::endl;
        return fail(std::string("GPU engine unavailable: ") + me_last_error(nullptr));
    }
    // The initial-matrix evaluation on one GPU is ONE library call (processOneCall: me_run_suite_from): without down-sampling it
    // starts from the host clouds as they were read — the uploads are part of the call's two-lane schedule.
    const bool one_call = param_.evaluate_using_initial_ && !comm_;
    if (one_call && !(param_.downsample_size > 0)) {
        file_result << std::fixed << std::setprecision(15) << "Estimated-Ground Truth point count: " << map_3d_->size() << " / "
                    << gt_3d_->size() << std::endl;
        if (param_.enable_debug)
            std::cout << "INFO: Loaded point clouds: " << map_3d_->size() << " points (Map), " << gt_3d_->size()
                      << " points (Ground Truth)." << std::endl;
        return processOneCall(true, tic_toc.toc());
    }
    if (me_upload_cloud(ctx_, ME_SLOT_GT, gt_3d_->points_.data(), (int64_t) gt_3d_->size(), nullptr, param_.nn_radius_) != ME_OK ||
        me_upload_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data(), (int64_t) map_3d_->size(), nullptr, param_.nn_radius_) != ME_OK)
        return fail(me_last_error(ctx_));
    if (gt_normals.size() == gt_3d_->points_.size() && !gt_normals.empty() &&
        me_set_normals(ctx_, ME_SLOT_GT, gt_normals.data()) != ME_OK)
        return fail(me_last_error(ctx_));
    if (map_normals.size() == map_3d_->points_.size() && !map_normals.empty() &&
        me_set_normals(ctx_, ME_SLOT_EST, map_normals.data()) != ME_OK)
        return fail(me_last_error(ctx_));
    // map_3d_ = map_3d_->VoxelDownSample(downsample_size) (:38-39), on the device (normals are averaged with the points)
    if (param_.downsample_size > 0) {
        int64_t ne = 0, ng = 0;
        if (me_voxel_downsample(ctx_, ME_SLOT_EST, param_.downsample_size, &ne) != ME_OK ||
            me_voxel_downsample(ctx_, ME_SLOT_GT, param_.downsample_size, &ng) != ME_OK)
            return fail(me_last_error(ctx_));
        map_3d_->points_.resize((size_t) ne * 3);
        gt_3d_->points_.resize((size_t) ng * 3);
        if (me_download_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data()) != ME_OK ||
            me_download_cloud(ctx_, ME_SLOT_GT, gt_3d_->points_.data()) != ME_OK)
            return fail(me_last_error(ctx_));
    }
    file_result << std::fixed << std::setprecision(15) << "Estimated-Ground Truth point count: " << map_3d_->size() << " / "
                << gt_3d_->size() << std::endl;
    if (param_.enable_debug)
        std::cout << "INFO: Loaded point clouds: " << map_3d_->size() << " points (Map), " << gt_3d_->size()
                  << " points (Ground Truth)." << std::endl;
    if (comm_) return processDist(tic_toc.toc());  // num_gpus > 1 (map_eval_dist.cpp)
    if (one_call) return processOneCall(false, tic_toc.toc());  // (the down-sampled clouds are resident)
    t1 = tic_toc.toc();
    // The reference computes MME on the map as loaded (:56) and transforms it afterwards, inside
    // calculateMetricsWithInitialMatrix (:1206): same order here (me_transform_cloud below), skipped for an identity matrix.
    const double *T = param_.evaluate_using_initial_ ? param_.initial_matrix_.data() : nullptr;
    bool identity = true;
    for (int i = 0; i < 16 && T; ++i) identity = identity && (T[i] == ((i % 5 == 0) ? 1.0 : 0.0));

    if (param_.evaluate_mme_) {
        if (param_.enable_debug) std::cout << "INFO: Starting MME calculation..." << std::endl;
        computeMME(*map_3d_, *gt_3d_);
        if (!last_error.empty()) return -1;
        if (param_.save_immediate_result_) saveMmeResults();
        t2 = tic_toc.toc();
        if (param_.enable_debug) std::cout << "INFO: MME calculation completed in: " << (t2 - t1) / 1000.0 << " seconds." << std::endl;
    } else {
        t2 = t1;
    }

    if (param_.evaluate_using_initial_) {
        if (param_.enable_debug) std::cout << "INFO: Using initial matrix without registration." << std::endl;
        if (T && !identity) {  // *map_3d_ = map_3d_->Transform(param_.initial_matrix_) (:1206)
            if (me_transform_cloud(ctx_, ME_SLOT_EST, T) != ME_OK ||
                me_download_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data()) != ME_OK)
                return fail(me_last_error(ctx_));
        }
        calculateMetricsWithInitialMatrix();
        if (!last_error.empty()) return -1;
    } else {
        // performRegistration (map_eval.cpp:191-237): point-to-point (0), point-to-plane (1) and generalized ICP (2) all run
        // on the device-side correspondence + reduction step; the small solve per iteration is done here.
        if (param_.evaluation_method_ < 0 || param_.evaluation_method_ > 2)
            return fail("Invalid registration type specified");  // (:1385-1387)
        if (performRegistration() != 0) return -1;
    }
    if (param_.evaluate_using_initial_) t5 = t4 = t3 = tic_toc.toc();

    calculateVMD();
    if (!last_error.empty()) return -1;
    if (param_.enable_debug) std::cout << "INFO: VMD calculation completed." << std::endl;
    if (param_.save_immediate_result_) saveRegistrationResults();
    if (param_.enable_debug) std::cout << "INFO: Results saved successfully." << std::endl;
    return 0;
}

namespace {

// symmetric n x n Jacobi eigen-decomposition (n <= 4): eigenvalues in d, eigenvectors in the columns of V
void jacobi_sym(int n, double *a, double *d, double *V) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[n * i + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a[n * p + q] * a[n * p + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[n * p + q];
                if (apq == 0.0) continue;
                const double theta = (a[n * q + q] - a[n * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[n * k + p], akq = a[n * k + q];
                    a[n * k + p] = c * akp - sn * akq;
                    a[n * k + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[n * p + k], aqk = a[n * q + k];
                    a[n * p + k] = c * apk - sn * aqk;
                    a[n * q + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[n * k + p], vkq = V[n * k + q];
                    V[n * k + p] = c * vkp - sn * vkq;
                    V[n * k + q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) d[i] = a[n * i + i];
}

// Optimal rigid update (row-major 4x4, absolute coordinates) from the me_icp_sums block: Horn's closed form, which gives
// the same rotation as Eigen::umeyama without scaling (TransformationEstimationPointToPoint [Open3D, upstream]).
void kabsch_from_sums(const me_icp_sums &s, double T[16]) {
    const double n = (double) s.n_corr;
    double pb[3], qb[3], S[9];
    for (int k = 0; k < 3; ++k) {
        pb[k] = s.sum_p[k] / n;
        qb[k] = s.sum_q[k] / n;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[3 * r + c] = s.sum_pq[3 * r + c] - n * pb[r] * qb[c];  // sum (p-pb)(q-qb)^T
    double N[16] = {S[0] + S[4] + S[8], S[5] - S[7],        S[6] - S[2],         S[1] - S[3],
                    S[5] - S[7],        S[0] - S[4] - S[8], S[1] + S[3],         S[6] + S[2],
                    S[6] - S[2],        S[1] + S[3],        -S[0] + S[4] - S[8], S[5] + S[7],
                    S[1] - S[3],        S[6] + S[2],        S[5] + S[7],         -S[0] - S[4] + S[8]};
    double d[4], V[16];
    jacobi_sym(4, N, d, V);
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (d[i] > d[best]) best = i;
    const double w = V[best], x = V[4 + best], y = V[8 + best], z = V[12 + best];
    const double R[9] = {w * w + x * x - y * y - z * z, 2 * (x * y - w * z),           2 * (x * z + w * y),
                         2 * (x * y + w * z),           w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                         2 * (x * z - w * y),           2 * (y * z + w * x),           w * w - x * x - y * y + z * z};
    // x -> o + R (x - o - pb) + qb
    for (int i = 0; i < 16; ++i) T[i] = (i == 15) ? 1.0 : 0.0;
    for (int r = 0; r < 3; ++r) {
        double t = s.origin[r] + qb[r];
        for (int c = 0; c < 3; ++c) {
            T[4 * r + c] = R[3 * r + c];
            t -= R[3 * r + c] * (s.origin[c] + pb[c]);
        }
        T[4 * r + 3] = t;
    }
}

// utility::SolveJacobianSystemAndObtainExtrinsicMatrix [Open3D, upstream]: x = solve(JTJ, -JTr) (LDLT there, Gaussian
// elimination with partial pivoting here), then TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x[3..5].
// A singular system leaves the identity (ComputeTransformation's failure value).
void lsq_update(const me_icp_lsq &q, double T[16]) {
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    double A[6][7];
    for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) A[r][c] = q.JTJ[6 * r + c];
        A[r][6] = -q.JTr[r];
    }
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        for (int r = k + 1; r < 6; ++r)
            if (std::fabs(A[r][k]) > std::fabs(A[piv][k])) piv = r;
        if (A[piv][k] == 0.0 || !std::isfinite(A[piv][k])) return;
        if (piv != k)
            for (int c = 0; c < 7; ++c) std::swap(A[piv][c], A[k][c]);
        for (int r = k + 1; r < 6; ++r) {
            const double f = A[r][k] / A[k][k];
            for (int c = k; c < 7; ++c) A[r][c] -= f * A[k][c];
        }
    }
    double x[6];
    for (int k = 5; k >= 0; --k) {
        double acc = A[k][6];
        for (int c = k + 1; c < 6; ++c) acc -= A[k][c] * x[c];
        x[k] = acc / A[k][k];
        if (!std::isfinite(x[k])) return;
    }
    const double ca = std::cos(x[0]), sa = std::sin(x[0]), cb = std::cos(x[1]), sb = std::sin(x[1]), cg = std::cos(x[2]),
                 sg = std::sin(x[2]);
    // Rz(g) * Ry(b) * Rx(a)
    T[0] = cg * cb;
    T[1] = cg * sb * sa - sg * ca;
    T[2] = cg * sb * ca + sg * sa;
    T[4] = sg * cb;
    T[5] = sg * sb * sa + cg * ca;
    T[6] = sg * sb * ca - cg * sa;
    T[8] = -sb;
    T[9] = cb * sa;
    T[10] = cb * ca;
    T[3] = x[3];
    T[7] = x[4];
    T[11] = x[5];
}

void matmul4(const double A[16], const double B[16], double C[16]) {
    double out[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += A[4 * r + k] * B[4 * k + c];
            out[4 * r + c] = acc;
        }
    for (int i = 0; i < 16; ++i) C[i] = out[i];
}

}  // namespace

int MapEval::performRegistration(bool metrics) {
    // RegistrationICP(*map_3d_, *gt_3d_, icp_max_distance_, initial_matrix_, PointToPoint, ICPConvergenceCriteria())
    // (map_eval.cpp:1369-1371): relative_fitness = relative_rmse = 1e-6, max_iteration = 30 [Open3D defaults, upstream]
    // Multi-GPU (comm_): both clouds are resident on every rank, the correspondence searches are sharded over the ranks
    // (me_set_shard: rank r searches the r-th share of the curve-sorted map), the step's additive sums are all-reduced, every rank
    // solves the same small system and moves its copy of the map: the iterates are those of the single-GPU loop up to the order
    // in which the sums are added.
    TicToc tic_toc;
    const bool root = param_.dist_rank == 0;
    if (comm_ && me_set_shard(ctx_, comm_->rank, comm_->world) != ME_OK) return fail(me_last_error(ctx_));
    for (int i = 0; i < 16; ++i) trans[i] = param_.initial_matrix_[i];
    bool identity = true;
    for (int i = 0; i < 16; ++i) identity = identity && (trans[i] == ((i % 5 == 0) ? 1.0 : 0.0));
    const int method = param_.evaluation_method_;
    if (method == 2) {
        // RegistrationGeneralizedICP (:1378-1384): InitializePointCloudForGeneralizedICP(epsilon = 1e-3) on both clouds as
        // loaded (normals from the 20 nearest neighbours where a cloud has none); they rotate with the map from here on
        if (me_gicp_covariances(ctx_, ME_SLOT_EST, 1e-3, nullptr) != ME_OK || me_gicp_covariances(ctx_, ME_SLOT_GT, 1e-3, nullptr) != ME_OK)
            return fail(me_last_error(ctx_));
    }
    if (!identity && me_transform_cloud(ctx_, ME_SLOT_EST, trans.data()) != ME_OK) return fail(me_last_error(ctx_));
    me_icp_sums s;
    me_icp_lsq q;
    auto evaluate = [&](double &fit, double &rmse) -> bool {
        if (me_nn1(ctx_, ME_SLOT_EST, ME_SLOT_GT, nullptr, nullptr) != ME_OK) return false;
        if (method == 0) {
            if (me_icp_p2p_sums(ctx_, ME_SLOT_EST, param_.icp_max_distance_, &s) != ME_OK) return false;
        } else {
            // TransformationEstimationPointToPlane (:1373-1377) needs normals on the target, as in Open3D
            if (me_icp_lsq_sums(ctx_, ME_SLOT_EST, method == 1 ? ME_ICP_POINT_TO_PLANE : ME_ICP_GENERALIZED, param_.icp_max_distance_, &q) != ME_OK)
                return false;
            s.n_corr = q.n_corr;
            s.n_source = q.n_source;
            s.sum_d2 = q.sum_d2;
        }
        if (comm_ && reduceIcp(s, q, method) != 0) return false;
        fit = s.n_source ? (double) s.n_corr / (double) s.n_source : 0.0;
        rmse = s.n_corr ? std::sqrt(s.sum_d2 / (double) s.n_corr) : 0.0;
        return true;
    };
    double fit = 0, rmse = 0;
    if (!evaluate(fit, rmse)) return fail(me_last_error(ctx_));
    int it = 0;
    for (it = 1; it <= 30; ++it) {
        if (method == 0 ? s.n_corr < 3 : s.n_corr == 0) break;
        double upd[16];
        if (method == 0) kabsch_from_sums(s, upd);
        else lsq_update(q, upd);
        matmul4(upd, trans.data(), trans.data());
        if (me_transform_cloud(ctx_, ME_SLOT_EST, upd) != ME_OK) return fail(me_last_error(ctx_));
        const double pf = fit, pr = rmse;
        if (!evaluate(fit, rmse)) return fail(me_last_error(ctx_));
        if (std::fabs(pf - fit) < 1e-6 && std::fabs(pr - rmse) < 1e-6) break;
    }
    t3 = 0;  // no mesh stage
    t4 = tic_toc.toc();
    if (comm_ && me_set_shard(ctx_, 0, 1) != ME_OK) return fail(me_last_error(ctx_));
    if (me_download_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data()) != ME_OK)  // *map_3d_ = map_3d_->Transform(trans) (:1392)
        return fail(me_last_error(ctx_));
    if (root) {
        std::cout << "INFO: ICP registration time: " << t4 / 1000.0 << " [s]" << std::endl;
        std::cout << "INFO: Aligned transformation: \n";
        for (int r = 0; r < 4; ++r)
            std::cout << trans[4 * r] << " " << trans[4 * r + 1] << " " << trans[4 * r + 2] << " " << trans[4 * r + 3] << std::endl;
        std::cout << "INFO: ICP overlap ratio: " << fit << std::endl;
        std::cout << "INFO: ICP correspondences RMSE: " << rmse << std::endl;
        std::cout << "INFO: ICP correspondences size: " << s.n_corr << std::endl;
    }
    // "Aligned cloud:" / "Aligned results:" lines (map_eval.cpp:223-225)
    // (`file_result << matrix`: Eigen's default IOFormat pads every coefficient to the width of the widest one of the WHOLE
    //  matrix, one space between columns, one row per line)
    file_result << std::fixed << std::setprecision(5) << "Aligned cloud: " << eigen_matrix4(trans.data(), 5) << std::endl;
    file_result << std::fixed << std::setprecision(5) << "Aligned results: " << fit << " " << s.n_corr << std::endl;
    if (metrics) calculateMetrics();
    t5 = tic_toc.toc();
    return last_error.empty() ? 0 : -1;
}

void MapEval::calculateMetrics() {
    // map_eval.cpp:1147-1202: statistics on ICP's final correspondence set (est -> gt, d2 < max^2), then
    // EvaluateRegistration(gt -> map, max) (:1168), cd_vec (:1171) and the full Chamfer distance (:1194).
    TicToc tt;
    me_nn_stats_out eg, ge;
    if (me_nn1(ctx_, ME_SLOT_EST, ME_SLOT_GT, nullptr, nullptr) != ME_OK ||
        me_nn_stats(ctx_, ME_SLOT_EST, param_.icp_max_distance_, ME_GATE_LT_SQUARED, param_.trunc_dist_.data(), &eg) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    t_acc = tt.toc() / 1000.0;
    if (me_nn1(ctx_, ME_SLOT_GT, ME_SLOT_EST, nullptr, nullptr) != ME_OK ||
        me_nn_stats(ctx_, ME_SLOT_GT, param_.icp_max_distance_, ME_GATE_LT_SQUARED, param_.trunc_dist_.data(), &ge) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    finishRegistrationMetrics(eg, ge, t_acc);
    t_fcd += tt.toc() / 1000.0 - t_acc;
}

// the tail of calculateMetrics (map_eval.cpp:1171-1201): result vectors, cd_vec, the full Chamfer distance
void MapEval::finishRegistrationMetrics(const me_nn_stats_out &eg, const me_nn_stats_out &ge, double t_acc_s) {
    t_acc = t_acc_s;
    push_results(est_gt_results, eg);
    push_results(gt_est_results, ge);
    for (int i = 0; i < 5; ++i) cd_vec[i] = est_gt_results[1][i] + gt_est_results[1][i];  // (:1171)
    TicToc t1_;
    full_chamfer_dist = eg.mean_nn_dist + ge.mean_nn_dist;  // computeChamferDistance (:1194, :1429): same two searches
    t_fcd = t1_.toc() / 1000.0;
    if (param_.dist_rank > 0) return;
    std::cout << "INFO: RMSE/AC: " << eigen_row(est_gt_results[1], 6) << std::endl;
    std::cout << "INFO: Fitness/Overlap: " << eigen_row(est_gt_results[2], 6) << std::endl;
    std::cout << "INFO: Full Chamfer distance: " << full_chamfer_dist << std::endl;
}

// open3d ColorToUint8 + the packed-float rgb column of open3d::io::WritePointCloud(.pcd)
static std::vector<float> pack_rgb(const std::vector<double> &rgb) {
    std::vector<float> out(rgb.size() / 3);
    for (size_t i = 0; i < out.size(); ++i) {
        uint32_t c[3];
        for (int k = 0; k < 3; ++k) c[k] = (uint32_t) std::round(std::min(1.0, std::max(0.0, rgb[3 * i + k])) * 255.0);
        const uint32_t packed = (c[0] << 16) | (c[1] << 8) | c[2];
        std::memcpy(&out[i], &packed, 4);
    }
    return out;
}

// ColorPointCloudByMME(cloud, entropies) (map_eval.cpp:686-735) from the entropies the last me_mme left on the device
bool MapEval::renderEntropy(int slot, std::vector<double> &xyz, std::vector<double> &rgb, bool want_points) {
    int64_t m = 0;
    if (me_render_entropy(ctx_, slot, nullptr, nullptr, 0, &m, &min_abs_entropy, &max_abs_entropy) != ME_OK) {
        fail(me_last_error(ctx_));
        return false;
    }
    xyz.clear();
    rgb.clear();
    if (m == 0) {
        // no point had enough neighbours (sparse cloud / small nn_radius): the entropy range is undefined (the reference
        // reads min/max of an empty set there).  Say so, and report NaN rather than the +-inf of an empty reduction.
        std::cerr << "WARNING: no valid entropy value (0 points with enough neighbours within nn_radius): MME range is undefined, "
                     "no entropy map is written" << std::endl;
        min_abs_entropy = max_abs_entropy = std::nan("");
        return true;
    }
    if (!want_points) return true;
    xyz.resize((size_t) m * 3);
    rgb.resize((size_t) m * 3);
    if (me_render_entropy(ctx_, slot, xyz.data(), rgb.data(), m, &m, &min_abs_entropy, &max_abs_entropy) != ME_OK) {
        fail(me_last_error(ctx_));
        return false;
    }
    return true;
}

// MapEval::process()'s metric phase (map_eval.cpp:52-85: computeMME :56, calculateMetricsWithInitialMatrix :76, calculateVMD :85)
// as ONE call into the library, made from this one thread (process() is single-threaded, :4): me_run_suite_from runs the stages
// on two lanes (its second lane is a thread inside the library, csrc/me_suite.hip).  The reference's member functions keep their
// roles below as CONSUMERS of what the call left on the device: entropies / colour maps, result vectors, the voxel files.
// (the clouds are std::vectors: pageable.  The library moves them through its own pinned staging buffers at the link's rate;
//  ME_SUITE_PIN_HOST_INPUT — page-locking the vectors in place for the call — is the alternative for callers short of host threads)
static const int kSuiteFlags = ME_SUITE_OVERLAP;
int MapEval::processOneCall(bool from_host, double t_loaded) {
    t1 = t_loaded;
    TicToc clock;
    me_suite_params sp{};
    sp.icp_max_distance = param_.icp_max_distance_;
    sp.gate_mode = ME_GATE_LE_UNSQUARED;  // d2 <= icp_max_distance (:1219, sic)
    for (int k = 0; k < 5; ++k) sp.trunc[k] = param_.trunc_dist_[k];
    sp.nn_radius = param_.nn_radius_;
    sp.vmd_voxel_size = param_.vmd_voxel_size_;
    sp.evaluate_mme = param_.evaluate_mme_ ? 1 : 0;
    sp.evaluate_gt_mme = param_.evaluate_gt_mme_ ? 1 : 0;
    sp.min_pts = 100;
    sp.scs_radius = 5;
    me_suite_out so;
    const double *T = param_.initial_matrix_.data();
    bool identity = true;
    for (int i = 0; i < 16; ++i) identity = identity && (T[i] == ((i % 5 == 0) ? 1.0 : 0.0));
    const int rc = from_host ? me_run_suite_from(ctx_, map_3d_->points_.data(), (int64_t) map_3d_->size(), gt_3d_->points_.data(),
                                                 (int64_t) gt_3d_->size(), T, &sp, kSuiteFlags, &so)
                             : me_run_suite_from(ctx_, nullptr, 0, nullptr, 0, T, &sp, ME_SUITE_OVERLAP, &so);
    if (rc != ME_OK) return fail(me_last_error(ctx_));
    const double suite_ms = clock.toc();
    std::cout << std::fixed << std::setprecision(3) << "INFO: metric phase, one me_run_suite_from call (two lanes): " << suite_ms
              << " ms for " << map_3d_->size() << " + " << gt_3d_->size() << " points [index " << so.stage_ms[0] << ", mme " << so.stage_ms[4]
              << " + " << so.stage_ms[5] << ", nn " << so.stage_ms[1] << " + " << so.stage_ms[2] << ", stats " << so.stage_ms[3]
              << ", awd/scs " << so.stage_ms[6] << "]" << std::endl;
    std::cout.unsetf(std::ios::floatfield);
    std::cout << std::setprecision(6);
    // ---- computeMME's members (:149-189) ----
    if (param_.evaluate_mme_) {
        mme_est = so.mme_est;
        mme_gt = so.mme_gt;
        if (param_.save_immediate_result_) {  // the per-point arrays are fetched only when something is written from them
            est_entropies.assign(map_3d_->size(), 0.0);
            valid_entropy_points.assign(map_3d_->size(), 0);
            if (me_mme_fetch(ctx_, ME_SLOT_EST, est_entropies.data(), valid_entropy_points.data()) != ME_OK) return fail(me_last_error(ctx_));
        }
        if (!renderEntropy(ME_SLOT_EST, map_entropy_xyz, map_entropy_rgb, param_.save_immediate_result_)) return -1;
        if (!identity && param_.save_immediate_result_) {
            // map_3d_entropy is built from the map AS LOADED (:179, before :1206): the colours above are those of the untransformed
            // map's entropies, the coordinates are taken from the host's copy, which is still untransformed here
            size_t k = 0;
            for (size_t i = 0; i < valid_entropy_points.size() && 3 * k + 2 < map_entropy_xyz.size(); ++i)
                if (valid_entropy_points[i]) {
                    for (int d = 0; d < 3; ++d) map_entropy_xyz[3 * k + d] = map_3d_->points_[3 * i + d];
                    ++k;
                }
        }
        const int64_t nv = so.mme_est_valid;
        if (param_.enable_debug)
            std::cout << "TBB MME Valid_points " << nv * 100.0 / (double) map_3d_->size() << "% " << nv << " " << map_3d_->size() << std::endl;
        if (nv * 100.0 / (double) map_3d_->size() < 0.6) std::cerr << "valid points is too small, please check the input point cloud" << std::endl;
        if (param_.evaluate_gt_mme_) {
            if (param_.save_immediate_result_) {
                gt_entropies.assign(gt_3d_->size(), 0.0);
                if (me_mme_fetch(ctx_, ME_SLOT_GT, gt_entropies.data(), nullptr) != ME_OK) return fail(me_last_error(ctx_));
            }
            if (!renderEntropy(ME_SLOT_GT, gt_entropy_xyz, gt_entropy_rgb, param_.save_immediate_result_)) return -1;
            std::cout << "MME EST-GT: " << mme_est << " " << mme_gt << std::endl;
        } else {
            std::cout << "MME EST: " << mme_est << std::endl;
        }
        if (param_.save_immediate_result_) saveMmeResults();
    }
    t2 = t1 + so.stage_ms[0] + so.stage_ms[4] + so.stage_ms[5];
    // ---- calculateMetricsWithInitialMatrix's members (:1204-1260) ----
    if (param_.enable_debug) std::cout << "INFO: Using initial matrix without registration." << std::endl;
    if (!identity && me_download_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data()) != ME_OK)  // *map_3d_ = map_3d_->Transform(..) (:1206)
        return fail(me_last_error(ctx_));
    finishInitialMatrixMetrics(so.est_gt, so.gt_est, so.stage_ms[1] / 1000.0);
    t_fcd += so.stage_ms[2] / 1000.0;  // the gt -> est search is what the full Chamfer distance adds (:1194)
    t5 = t4 = t3 = t2 + so.stage_ms[1] + so.stage_ms[2] + so.stage_ms[3];
    // ---- calculateVMD (:240-390): the voxel tables are cached on the clouds, AWD / CDF / SCS are O(voxels) ----
    calculateVMD();
    if (!last_error.empty()) return -1;
    if (param_.enable_debug) std::cout << "INFO: VMD calculation completed." << std::endl;
    if (param_.save_immediate_result_) saveRegistrationResults();
    if (param_.enable_debug) std::cout << "INFO: Results saved successfully." << std::endl;
    return 0;
}

void MapEval::computeMME(PointCloud &cloud, PointCloud &gt) {
    // est: ComputeMeanMapEntropyUsingNormal[TBB] k >= 10 (map_eval.cpp:1675); gt: ComputeMeanMapEntropy k >= 5 (:1458)
    est_entropies.assign(cloud.size(), 0.0);
    valid_entropy_points.assign(cloud.size(), 0);
    double s = 0;
    int64_t nv = 0;
    if (me_mme(ctx_, ME_SLOT_EST, param_.nn_radius_, 10, est_entropies.data(), valid_entropy_points.data(), &s, &nv) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    mme_est = nv > 0 ? s / (double) nv : 0.0;
    // map_3d_entropy = ColorPointCloudByMME(map_3d_, est_entropies) (:136, :179) — on the device, from the entropies it holds
    if (!renderEntropy(ME_SLOT_EST, map_entropy_xyz, map_entropy_rgb, param_.save_immediate_result_)) return;
    if (param_.enable_debug)
        std::cout << "TBB MME Valid_points " << nv * 100.0 / (double) cloud.size() << "% " << nv << " " << cloud.size() << std::endl;
    if (nv * 100.0 / (double) cloud.size() < 0.6) std::cerr << "valid points is too small, please check the input point cloud" << std::endl;
    if (param_.evaluate_gt_mme_) {
        gt_entropies.assign(gt.size(), 0.0);
        std::vector<uint8_t> gv(gt.size(), 0);
        if (me_mme(ctx_, ME_SLOT_GT, param_.nn_radius_, 5, gt_entropies.data(), gv.data(), &s, &nv) != ME_OK) {
            fail(me_last_error(ctx_));
            return;
        }
        mme_gt = nv > 0 ? s / (double) nv : 0.0;
        // gt_3d_entropy = ColorPointCloudByMME(gt_3d_, gt_entropies) (:139, :181); like the reference, this call overwrites
        // min/max_abs_entropy with the GT range (they are members there, :698-699)
        if (!renderEntropy(ME_SLOT_GT, gt_entropy_xyz, gt_entropy_rgb, param_.save_immediate_result_)) return;
        std::cout << "MME EST-GT: " << mme_est << " " << mme_gt << std::endl;
    } else {
        std::cout << "MME EST: " << mme_est << std::endl;
    }
}

void MapEval::calculateMetricsWithInitialMatrix() {
    TicToc tt;
    me_nn_stats_out eg, ge;
    // est -> gt: keep (i, nn) iff d2 <= icp_max_distance (:1215-1223, squared vs un-squared, sic)
    if (me_nn1(ctx_, ME_SLOT_EST, ME_SLOT_GT, nullptr, nullptr) != ME_OK ||
        me_nn_stats(ctx_, ME_SLOT_EST, param_.icp_max_distance_, ME_GATE_LE_UNSQUARED, param_.trunc_dist_.data(), &eg) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    t_acc = tt.toc() / 1000.0;
    // gt -> est (:1226-1236): the intended (gt_i, map_nn) pairing — the reference stores the pair swapped and then indexes
    // the wrong clouds (undefined behaviour when N_e != N_g); see DESIGN.md "deviations".
    if (me_nn1(ctx_, ME_SLOT_GT, ME_SLOT_EST, nullptr, nullptr) != ME_OK ||
        me_nn_stats(ctx_, ME_SLOT_GT, param_.icp_max_distance_, ME_GATE_LE_UNSQUARED, param_.trunc_dist_.data(), &ge) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    const double t_both = tt.toc() / 1000.0;
    finishInitialMatrixMetrics(eg, ge, t_acc);
    t_fcd += t_both - t_acc;  // the gt -> est search is what the full Chamfer distance adds (:1194)
}

void MapEval::finishInitialMatrixMetrics(const me_nn_stats_out &eg, const me_nn_stats_out &ge, double t_acc_s) {
    TicToc tt;
    t_acc = t_acc_s;
    push_results(est_gt_results, eg);
    push_results(gt_est_results, ge);
    for (int i = 0; i < 5; ++i) {
        cd_vec[i] = est_gt_results[1][i] + gt_est_results[1][i];  // (:1245)
        const double overlap_ratio = est_gt_results[2][i], rmse = est_gt_results[1][i];
        f1_vec[i] = 2 * overlap_ratio * rmse / (overlap_ratio + rmse);  // (:1249, sic)
        const int num_intersection = (int) est_gt_results[4][i];
        const long long num_union = (long long) map_3d_->size() + (long long) gt_3d_->size() - num_intersection;
        iou_vec[i] = (double) num_intersection / (double) num_union;  // (:1250-1252)
    }
    // FULL CD: the reference never computes it on this path (stays 0.0); it is free here (same two searches).
    full_chamfer_dist = param_.strict_reference ? 0.0 : (eg.mean_nn_dist + ge.mean_nn_dist);  // (:1429)
    t_fcd = tt.toc() / 1000.0;
    if (param_.dist_rank > 0) return;
    std::cout << "INFO: Chamfer Distance: " << eigen_row(cd_vec, 6) << std::endl;
    std::cout << "INFO: F1 Score: " << eigen_row(f1_vec, 6) << std::endl;
    std::cout << "INFO: est-gt MME: " << mme_est << " " << mme_gt << std::endl;
    std::cout << "INFO: IoU: " << eigen_row(iou_vec, 6) << std::endl;
}

double MapEval::computeChamferDistance() {
    double cd = 0;
    if (me_chamfer(ctx_, &cd) != ME_OK) fail(me_last_error(ctx_));
    return cd;
}

void MapEval::calculateVMD(bool tables_ready, bool write_files) {
    TicToc ticToc;
    int64_t nv = 0;
    // buildVoxelMap(gt), buildVoxelMap(est), updateVoxelMap (:248-252); tables_ready: the merged tables of a multi-GPU run
    if (!tables_ready)
    if (me_voxel_gaussians(ctx_, ME_SLOT_GT, param_.vmd_voxel_size_, nullptr, nullptr, nullptr, nullptr, nullptr, &nv) != ME_OK ||
        me_voxel_gaussians(ctx_, ME_SLOT_EST, param_.vmd_voxel_size_, nullptr, nullptr, nullptr, nullptr, nullptr, &nv) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    t_v = ticToc.toc();
    int64_t n_rows = 0, counts[3] = {0, 0, 0};
    if (me_awd_scs(ctx_, param_.vmd_voxel_size_, 100, 5, nullptr, nullptr, &n_rows, &vmd, &scs_overall, counts) != ME_OK) {
        fail(me_last_error(ctx_));
        return;
    }
    if (!write_files) return;  // (ranks > 0 of a multi-GPU run hold the same scalars and write nothing)
    std::cout << "Update active/old/new voxel num: " << counts[0] << " " << counts[1] << " " << counts[2] << std::endl;
    std::vector<double> rows((size_t) n_rows * 27), ws((size_t) n_rows);
    if (n_rows > 0) {
        int64_t cap = n_rows;
        if (me_awd_scs(ctx_, param_.vmd_voxel_size_, 100, 5, rows.data(), ws.data(), &cap, &vmd, &scs_overall, counts) != ME_OK) {
            fail(me_last_error(ctx_));
            return;
        }
    }
    t_vmd = ticToc.toc();
    // voxel_errors.txt: 27 columns, default ostream precision (:292-302); rows in ascending voxel-index order
    std::ofstream output_file(results_subfolder + "voxel_errors.txt");
    if (!output_file.is_open()) {
        std::cerr << "ERROR: Failed to open voxel error output file." << std::endl;
        return;
    }
    for (int64_t r = 0; r < n_rows; ++r) {
        const double *p = rows.data() + 27 * r;
        for (int c = 0; c < 27; ++c) {
            if (c == 10 || c == 11) output_file << (long long) p[c];
            else output_file << p[c];
            output_file << (c == 26 ? "" : " ");
        }
        output_file << std::endl;
    }
    output_file.close();
    std::cout << "INFO: Calculated VMD: " << vmd << std::endl;
    // voxel_wasserstein_cdf.txt (:330-341)
    std::ofstream cdf_file(results_subfolder + "voxel_wasserstein_cdf.txt");
    if (!cdf_file.is_open()) {
        std::cerr << "ERROR: Failed to open CDF output file." << std::endl;
        return;
    }
    for (size_t i = 0; i < ws.size(); ++i) cdf_file << ws[i] << " " << static_cast<double>(i + 1) / ws.size() << std::endl;
    cdf_file.close();
    t_cdf = ticToc.toc();
    t_scs = t_cdf;  // SCS ran inside me_awd_scs
    std::cout << "INFO: Spatial Consistency Score (SCS): " << scs_overall << std::endl;
}

void MapEval::saveMmeResults() {
    if (!param_.evaluate_mme_) return;
    file_result << std::fixed << std::setprecision(5) << "MME: " << mme_est << " " << mme_gt << " " << min_abs_entropy << " "
                << max_abs_entropy << std::endl;  // (:395-396)
    // map_entropy.pcd / gt_entropy.pcd (:404, :412): valid points + Jet colour of the log-mapped entropy
    if (!map_entropy_xyz.empty()) {  // (nothing to draw when no point has a valid entropy; renderEntropy has said so)
        pcio::write_pcd(results_subfolder + "map_entropy.pcd", map_entropy_xyz.data(), map_entropy_xyz.size() / 3,
                        pack_rgb(map_entropy_rgb).data());
        std::cout << "INFO: Saved rendered entropy map to " << results_subfolder + "map_entropy.pcd" << std::endl;
    }
    if (param_.evaluate_gt_mme_ && !gt_entropy_xyz.empty()) {
        pcio::write_pcd(results_subfolder + "gt_entropy.pcd", gt_entropy_xyz.data(), gt_entropy_xyz.size() / 3,
                        pack_rgb(gt_entropy_rgb).data());
        std::cout << "INFO: Saved rendered entropy ground truth map to " << results_subfolder + "gt_entropy.pcd" << std::endl;
    }
    // (extra) the raw per-point entropies, so that nothing is lost to the colour map
    std::ofstream e(results_subfolder + "map_entropy.txt");
    for (size_t i = 0; i < est_entropies.size(); ++i) e << est_entropies[i] << " " << (int) valid_entropy_points[i] << "\n";
}

void MapEval::saveRegistrationResults() {
    // identical lines and precisions to map_eval.cpp:439-476
    file_result << std::fixed << std::setprecision(15) << "RMSE/AC: " << eigen_row(est_gt_results.at(1), 15) << std::endl;
    file_result << std::fixed << std::setprecision(15) << "Comp: " << eigen_row(est_gt_results.at(2), 15) << std::endl;
    file_result << std::fixed << std::setprecision(5) << "FULL CD: " << full_chamfer_dist << std::endl;
    file_result << std::fixed << std::setprecision(5) << "VMD: " << vmd << std::endl;
    file_result << std::fixed << std::setprecision(5) << "SCS: " << scs_overall << std::endl;
    if (param_.evaluate_using_initial_)
        file_result << "Time load-MME-mesh-ICP-Metric-AC-FCD: " << t1 / 1000.0 << " " << (t2 - t1) / 1000.0 << " " << 0.0 << " "
                    << 0.0 << " " << (t5 - t2) / 1000.0 << " " << t_acc << " " << t_fcd << std::endl;
    else  // t3..t5 come from performRegistration's own clock (map_eval.cpp:192, :465-467)
        file_result << "Time load-MME-mesh-ICP-Metric-AC-FCD: " << t1 / 1000.0 << " " << (t2 - t1) / 1000.0 << " " << t3 / 1000.0
                    << " " << (t4 - t3) / 1000.0 << " " << (t5 - t4) / 1000.0 << " " << t_acc << " " << t_fcd << std::endl;
    file_result << "VMD Time voxelization-WD-CDF-SCS: " << t_v / 1000.0 << " " << (t_vmd - t_v) / 1000.0 << " "
                << (t_cdf - t_vmd) / 1000.0 << " " << (t_scs - t_cdf) / 1000.0 << std::endl;
    file_result << "AC+MME Time: " << t_acc + (t2 - t1) / 1000.0 << std::endl;
    file_result << "CD+MME Time: " << t_fcd + (t2 - t1) / 1000.0 << std::endl;
    file_result << "AWD+SCS Time: " << t_v / 1000.0 + (t_vmd - t_v) / 1000.0 + (t_scs - t_cdf) / 1000.0 << std::endl;
    file_result.close();
    if (param_.enable_debug) std::cout << "INFO: Results saved to " << results_subfolder + "map_results.txt" << std::endl;
    // raw_rendered_dis_map.pcd / inlier_rendered_dis_map.pcd (:485-495): the map coloured by min(d2, trunc[0]) / trunc[0]
    // (renderDistanceOnPointCloud, :586-607).  The reference runs a serial KD-tree pass for it; the squared distances of
    // the est -> gt search are still on the device.  Inlier cloud = corresponding_cloud_est, the gated rows (:1086-1087).
    {
        const size_t n = map_3d_->size();
        std::vector<double> rgb(n * 3);
        std::vector<uint8_t> inl(n);
        const int mode = param_.evaluate_using_initial_ ? ME_GATE_LE_UNSQUARED : ME_GATE_LT_SQUARED;
        if (me_render_distance(ctx_, ME_SLOT_EST, param_.trunc_dist_[0], param_.icp_max_distance_, mode, rgb.data(), inl.data()) !=
            ME_OK) {
            fail(me_last_error(ctx_));
            return;
        }
        pcio::write_pcd(results_subfolder + "raw_rendered_dis_map.pcd", map_3d_->points_.data(), n, pack_rgb(rgb).data());
        std::vector<double> ixyz, irgb;
        for (size_t i = 0; i < n; ++i)
            if (inl[i])
                for (int k = 0; k < 3; ++k) {
                    ixyz.push_back(map_3d_->points_[3 * i + k]);
                    irgb.push_back(rgb[3 * i + k]);
                }
        pcio::write_pcd(results_subfolder + "inlier_rendered_dis_map.pcd", ixyz.data(), ixyz.size() / 3, pack_rgb(irgb).data());
        if (param_.enable_debug)
            std::cout << "INFO: Saved raw / inlier distance error maps to " << results_subfolder << "{raw,inlier}_rendered_dis_map.pcd"
                      << std::endl;
    }
}
