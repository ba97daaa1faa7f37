// Compile-and-run check of the REFERENCE-SIDE binding shown in INTEGRATION.md section B (tests/test_integration_binding.py).
// binding_members.inc / binding_process.inc are the two code blocks of that section, extracted verbatim by the test; this file
// supplies what surrounds them in the reference: the members of MapEval they touch (map_eval/src/map_eval.h:60-116, :322-353)
// with the reference's names and types, over stand-in Open3D / Eigen headers (types only; neither library is installed here).
// Output: one line of scalars, compared by the GPU test with the Python face of the same library.
#include <open3d/Open3D.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <memory>
#include <random>
#include <vector>

using open3d::geometry::PointCloud;
typedef Eigen::Matrix<double, 5, 1> Vector5d;  // map_eval.h:48

struct Param {  // the hot-path fields of map_eval.h:60-116
    double icp_max_distance_ = 2.5;
    double nn_radius_ = 0.2;
    bool evaluate_gt_mme_ = true;
    Vector5d trunc_dist_;
    Eigen::Matrix4d initial_matrix_ = Eigen::Matrix4d::Identity();
    double vmd_voxel_size_ = 3.0;
    double downsample_size = 0.01;
};

class MapEval {
public:
    explicit MapEval(Param &p) : param_(p) {}
    ~MapEval() {
        if (gpu_) me_destroy(gpu_);
    }
    Param param_;
    std::shared_ptr<PointCloud> map_3d_{new PointCloud}, gt_3d_{new PointCloud};  // map_eval.h:322
    std::vector<Vector5d> est_gt_results, gt_est_results;                         // :328
    Vector5d cd_vec = Vector5d::Zero();                                           // :330
    double vmd = 0.0, full_chamfer_dist = 0.0, scs_overall = 0.0;                 // :333-337
    std::vector<double> est_entropies, gt_entropies;                              // :349
    double mme_est = 0.0, mme_gt = 0.0;                                           // :351
#include "binding_members.inc"
    int process() {
#include "binding_process.inc"
        std::printf("RESULT %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %lld\n", est_gt_results[1][0], est_gt_results[2][0],
                    gt_est_results[1][0], full_chamfer_dist, mme_est, mme_gt, vmd, scs_overall, (long long) n_rows);
        return 0;
    }
};

int main(int argc, char **argv) {
    // a small deterministic pair: a noisy 2 m x 2 m plane patch at ~2500 points / m^2 (the Python side rebuilds it from the file)
    const int n = argc > 1 ? std::atoi(argv[1]) : 20000;
    Param p;
    p.icp_max_distance_ = 1.0;
    p.nn_radius_ = 0.1;
    p.vmd_voxel_size_ = 0.5;
    const double tr[5] = {0.2, 0.1, 0.08, 0.05, 0.01};
    for (int k = 0; k < 5; ++k) p.trunc_dist_[k] = tr[k];
    p.initial_matrix_(0, 3) = 0.004;  // a small translation: exercises the column-major -> row-major hand-over
    p.initial_matrix_(1, 3) = -0.003;
    MapEval me(p);
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> u(0.0, 2.8), j(-1e-3, 1e-3);
    std::normal_distribution<double> nz(0.0, 0.01);
    for (int i = 0; i < n; ++i) {
        Eigen::Vector3d a, b;
        a[0] = u(g); a[1] = u(g); a[2] = j(g);
        b[0] = a[0] + nz(g); b[1] = a[1] + nz(g); b[2] = a[2] + nz(g);
        me.gt_3d_->points_.push_back(a);
        me.map_3d_->points_.push_back(b);
    }
    if (argc > 2) {  // dump the clouds for the Python side
        FILE *f = std::fopen(argv[2], "wb");
        std::fwrite(me.map_3d_->points_.data(), 24, n, f);
        std::fwrite(me.gt_3d_->points_.data(), 24, n, f);
        std::fclose(f);
    }
    return me.process() == 0 ? 0 : 3;
}
