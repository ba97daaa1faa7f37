// open3d/Open3D.h — FUNCTIONAL STAND-IN for the slice of Open3D (README: 0.15.1, CI: 0.17.0) that the reference's
// map_eval.cpp / voxel_calculator.cpp name.  TEST INFRASTRUCTURE ONLY (oracle/_ref): Open3D is absent from this image, so
// the reference's own sources are compiled, unmodified, against this header.  Everything here is the BUILDER's code:
//   * geometry::KDTreeFlann  — an exact KD-tree (leaf 15) with nanoflann's contract: SearchKNN returns SQUARED L2 distances
//     ((dx*dx + dy*dy) + dz*dz), ascending; SearchRadius(q, r) returns every point with d2 < r*r (strict), ascending.
//     Exact searches have ONE right answer (up to the order of equal distances), so results do not depend on the tree.
//   * PointCloud::Transform  — ((c0*x + c1*y) + c2*z) + c3 per row, divided by w (Open3D's homogeneous transform).
//   * PointCloud::VoxelDownSample — Open3D's voxel index floor((p - (min_bound - v/2)) / v), per-voxel mean in cloud order;
//     output in ascending voxel order (Open3D: its hash map's iteration order).
//   * pipelines::registration::EvaluateRegistration — 1-NN with d2 < max_distance^2 (SearchHybrid(q, max, 1)).
//   * visualization::ColorMapJet — Open3D's piecewise-linear jet.
//   * io::ReadPointCloudFromPCD / WritePointCloud — ascii / binary PCD with fp32 or fp64 x y z (what the tests write).
// NOT provided (they throw or do nothing; SURVEY section 2 marks them out of scope): Poisson meshing, normals, the ICP solvers,
// windows.
#pragma once
#include <Eigen/Core>

// (the real Open3D.h drags in most of the standard library; the reference relies on that for <numeric>, <random>, <mutex>)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace open3d {

namespace geometry {

class Geometry {
  public:
    virtual ~Geometry() {}
};

class KDTreeSearchParamHybrid {
  public:
    KDTreeSearchParamHybrid(double radius, int max_nn) : radius_(radius), max_nn_(max_nn) {}
    double radius_;
    int max_nn_;
};

class PointCloud : public Geometry {
  public:
    std::vector<Eigen::Vector3d> points_, normals_, colors_;

    bool IsEmpty() const { return points_.empty(); }
    bool HasNormals() const { return !points_.empty() && normals_.size() == points_.size(); }

    PointCloud &Transform(const Eigen::Matrix4d &T) {
        for (auto &p : points_) {
            double r[4];
            for (int i = 0; i < 4; ++i) r[i] = ((T(i, 0) * p(0) + T(i, 1) * p(1)) + T(i, 2) * p(2)) + T(i, 3);
            p = Eigen::Vector3d(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
        }
        return *this;
    }
    PointCloud &PaintUniformColor(const Eigen::Vector3d &c) {
        colors_.assign(points_.size(), c);
        return *this;
    }
    void EstimateNormals(const KDTreeSearchParamHybrid &) {}  // only behind eva_mesh (never true)

    std::shared_ptr<PointCloud> VoxelDownSample(double voxel_size) const {
        auto out = std::make_shared<PointCloud>();
        if (!(voxel_size > 0.0)) throw std::runtime_error("[VoxelDownSample] voxel_size <= 0.");
        if (points_.empty()) return out;
        Eigen::Vector3d mn = points_[0];
        for (const auto &p : points_)
            for (int d = 0; d < 3; ++d) mn(d) = std::min(mn(d), p(d));
        for (int d = 0; d < 3; ++d) mn(d) -= voxel_size * 0.5;
        struct Acc {
            double s[3] = {0, 0, 0};
            long c = 0;
        };
        std::map<std::tuple<int, int, int>, Acc> acc;
        for (const auto &p : points_) {
            const auto key = std::make_tuple((int) std::floor((p(0) - mn(0)) / voxel_size),
                                             (int) std::floor((p(1) - mn(1)) / voxel_size),
                                             (int) std::floor((p(2) - mn(2)) / voxel_size));
            Acc &a = acc[key];
            for (int d = 0; d < 3; ++d) a.s[d] += p(d);
            a.c++;
        }
        out->points_.reserve(acc.size());
        for (const auto &kv : acc)
            out->points_.emplace_back(kv.second.s[0] / (double) kv.second.c, kv.second.s[1] / (double) kv.second.c,
                                      kv.second.s[2] / (double) kv.second.c);
        return out;
    }
};

class TriangleMesh : public Geometry {
  public:
    std::vector<Eigen::Vector3d> vertices_, vertex_colors_, vertex_normals_;
    std::vector<Eigen::Vector3i> triangles_;
    static std::tuple<std::shared_ptr<TriangleMesh>, std::vector<double>> CreateFromPointCloudPoisson(const PointCloud &,
                                                                                                     size_t) {
        throw std::runtime_error("stand-in Open3D: Poisson reconstruction is not provided (out of scope)");
    }
    TriangleMesh &PaintUniformColor(const Eigen::Vector3d &c) {
        vertex_colors_.assign(vertices_.size(), c);
        return *this;
    }
    TriangleMesh &ComputeVertexNormals() { return *this; }
    void RemoveVerticesByMask(const std::vector<bool> &) {}
};

class KDTreeFlann {
    struct Node {
        int lo = 0, hi = 0;          // leaf: [lo, hi) into perm_
        int left = -1, right = -1;   // inner: children
        int dim = 0;
        double divlow = 0, divhigh = 0;
    };
    const Eigen::Vector3d *pts_ = nullptr;
    int n_ = 0;
    std::vector<int> perm_;
    std::vector<Node> nodes_;
    double lo_[3], hi_[3];

    double coord(int i, int d) const { return pts_[i](d); }
    int build(int lo, int hi) {
        const int id = (int) nodes_.size();
        nodes_.emplace_back();
        if (hi - lo <= 15) {
            nodes_[id].lo = lo;
            nodes_[id].hi = hi;
            return id;
        }
        double mn[3], mx[3];
        for (int d = 0; d < 3; ++d) mn[d] = mx[d] = coord(perm_[lo], d);
        for (int k = lo + 1; k < hi; ++k)
            for (int d = 0; d < 3; ++d) {
                const double v = coord(perm_[k], d);
                mn[d] = std::min(mn[d], v);
                mx[d] = std::max(mx[d], v);
            }
        int dim = 0;
        for (int d = 1; d < 3; ++d)
            if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
        const int mid = lo + (hi - lo) / 2;
        std::nth_element(perm_.begin() + lo, perm_.begin() + mid, perm_.begin() + hi,
                         [&](int a, int b) { return coord(a, dim) < coord(b, dim); });
        double dl = coord(perm_[lo], dim), dh = coord(perm_[mid], dim);
        for (int k = lo; k < mid; ++k) dl = std::max(dl, coord(perm_[k], dim));
        for (int k = mid; k < hi; ++k) dh = std::min(dh, coord(perm_[k], dim));
        const int l = build(lo, mid);
        const int r = build(mid, hi);
        Node &nd = nodes_[id];
        nd.left = l;
        nd.right = r;
        nd.dim = dim;
        nd.divlow = dl;
        nd.divhigh = dh;
        return id;
    }
    static double d2(const Eigen::Vector3d &a, const Eigen::Vector3d &b) {
        const double dx = a(0) - b(0), dy = a(1) - b(1), dz = a(2) - b(2);
        return (dx * dx + dy * dy) + dz * dz;
    }
    template <typename RS>
    void search(RS &rs, const Eigen::Vector3d &q, int id, double mind, double dists[3]) const {
        const Node &nd = nodes_[id];
        if (nd.left < 0) {
            for (int k = nd.lo; k < nd.hi; ++k) {
                const int i = perm_[k];
                const double d = d2(q, pts_[i]);
                if (d < rs.worst()) rs.add(d, i);
            }
            return;
        }
        const double v = q(nd.dim), diff1 = v - nd.divlow, diff2 = v - nd.divhigh;
        int best, other;
        double cut;
        if (diff1 + diff2 < 0) {
            best = nd.left;
            other = nd.right;
            cut = diff2 * diff2;
        } else {
            best = nd.right;
            other = nd.left;
            cut = diff1 * diff1;
        }
        search(rs, q, best, mind, dists);
        const double dst = dists[nd.dim];
        mind = mind + cut - dst;
        dists[nd.dim] = cut;
        if (mind <= rs.worst()) search(rs, q, other, mind, dists);
        dists[nd.dim] = dst;
    }
    template <typename RS>
    void run(RS &rs, const Eigen::Vector3d &q) const {
        if (n_ == 0) return;
        double dists[3], mind = 0;
        for (int d = 0; d < 3; ++d) {
            dists[d] = 0;
            if (q(d) < lo_[d]) dists[d] = (q(d) - lo_[d]) * (q(d) - lo_[d]);
            if (q(d) > hi_[d]) dists[d] = (q(d) - hi_[d]) * (q(d) - hi_[d]);
            mind += dists[d];
        }
        search(rs, q, 0, mind, dists);
    }
    struct KnnSet {
        int k, cnt = 0;
        std::vector<int> &idx;
        std::vector<double> &dist;
        KnnSet(int k_, std::vector<int> &i, std::vector<double> &d) : k(k_), idx(i), dist(d) {
            idx.assign(k, -1);
            dist.assign(k, std::numeric_limits<double>::max());
        }
        double worst() const { return dist[k - 1]; }
        void add(double d, int i) {
            int p = std::min(cnt, k - 1);
            while (p > 0 && dist[p - 1] > d) {
                dist[p] = dist[p - 1];
                idx[p] = idx[p - 1];
                --p;
            }
            dist[p] = d;
            idx[p] = i;
            if (cnt < k) ++cnt;
        }
    };
    struct RadiusSet {
        double r2;
        std::vector<std::pair<double, int>> found;
        double worst() const { return r2; }
        void add(double d, int i) { found.emplace_back(d, i); }
    };

  public:
    KDTreeFlann() {}
    explicit KDTreeFlann(const PointCloud &pc) { SetGeometry(pc); }
    bool SetGeometry(const PointCloud &pc) {
        pts_ = pc.points_.data();
        n_ = (int) pc.points_.size();
        perm_.resize(n_);
        for (int i = 0; i < n_; ++i) perm_[i] = i;
        nodes_.clear();
        if (n_ == 0) return false;
        nodes_.reserve((size_t) n_ / 4 + 16);
        for (int d = 0; d < 3; ++d) lo_[d] = hi_[d] = coord(0, d);
        for (int i = 1; i < n_; ++i)
            for (int d = 0; d < 3; ++d) {
                lo_[d] = std::min(lo_[d], coord(i, d));
                hi_[d] = std::max(hi_[d], coord(i, d));
            }
        build(0, n_);
        return true;
    }
    int SearchKNN(const Eigen::Vector3d &q, int knn, std::vector<int> &indices, std::vector<double> &distance2) const {
        if (n_ == 0 || knn <= 0) return -1;
        KnnSet rs(knn, indices, distance2);
        run(rs, q);
        indices.resize(rs.cnt);
        distance2.resize(rs.cnt);
        return rs.cnt;
    }
    int SearchRadius(const Eigen::Vector3d &q, double radius, std::vector<int> &indices,
                     std::vector<double> &distance2) const {
        if (n_ == 0) return -1;
        RadiusSet rs;
        rs.r2 = radius * radius;
        run(rs, q);
        std::sort(rs.found.begin(), rs.found.end());  // ascending by distance (equal distances: by index)
        indices.resize(rs.found.size());
        distance2.resize(rs.found.size());
        for (size_t k = 0; k < rs.found.size(); ++k) {
            indices[k] = rs.found[k].second;
            distance2[k] = rs.found[k].first;
        }
        return (int) rs.found.size();
    }
};

}  // namespace geometry

namespace io {

struct ReadPointCloudOption {
    ReadPointCloudOption(const std::string &format = "auto", bool remove_nan = false, bool remove_inf = false,
                         bool print_progress = false)
        : format_(format), remove_nan_(remove_nan), remove_inf_(remove_inf), print_progress_(print_progress) {}
    std::string format_;
    bool remove_nan_, remove_inf_, print_progress_;
};

// ascii / binary PCD with F4 or F8 fields; x y z are picked by name, every other field is skipped.
inline bool ReadPointCloudFromPCD(const std::string &path, geometry::PointCloud &pc, const ReadPointCloudOption &opt) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<std::string> fields;
    std::vector<int> sizes, counts;
    std::vector<char> types;
    long points = -1, width = 0, height = 1;
    std::string data, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key, tok;
        ss >> key;
        if (key == "FIELDS") while (ss >> tok) fields.push_back(tok);
        else if (key == "SIZE") while (ss >> tok) sizes.push_back(std::stoi(tok));
        else if (key == "TYPE") while (ss >> tok) types.push_back(tok[0]);
        else if (key == "COUNT") while (ss >> tok) counts.push_back(std::stoi(tok));
        else if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "POINTS") ss >> points;
        else if (key == "DATA") {
            ss >> data;
            break;
        }
    }
    if (points < 0) points = width * height;
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (fields.size() != sizes.size() || fields.size() != types.size()) return false;
    int off[3] = {-1, -1, -1}, fsz[3] = {0, 0, 0}, col[3] = {-1, -1, -1}, stride = 0, ncol = 0;
    for (size_t k = 0; k < fields.size(); ++k) {
        for (int d = 0; d < 3; ++d)
            if (fields[k] == std::string(1, "xyz"[d])) {
                off[d] = stride;
                fsz[d] = sizes[k];
                col[d] = ncol;
                if (types[k] != 'F') return false;
            }
        stride += sizes[k] * counts[k];
        ncol += counts[k];
    }
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) return false;
    pc.points_.clear();
    pc.points_.reserve(points);
    auto keep = [&](const Eigen::Vector3d &p) {
        for (int d = 0; d < 3; ++d) {
            if (opt.remove_nan_ && std::isnan(p(d))) return false;
            if (opt.remove_inf_ && std::isinf(p(d))) return false;
        }
        return true;
    };
    if (data == "binary") {
        std::vector<char> rec(stride);
        for (long i = 0; i < points; ++i) {
            if (!f.read(rec.data(), stride)) return false;
            Eigen::Vector3d p;
            for (int d = 0; d < 3; ++d) {
                if (fsz[d] == 8) {
                    double v;
                    std::memcpy(&v, rec.data() + off[d], 8);
                    p(d) = v;
                } else {
                    float v;
                    std::memcpy(&v, rec.data() + off[d], 4);
                    p(d) = v;
                }
            }
            if (keep(p)) pc.points_.push_back(p);
        }
    } else if (data == "ascii") {
        for (long i = 0; i < points; ++i) {
            if (!std::getline(f, line)) return false;
            std::istringstream ss(line);
            Eigen::Vector3d p;
            std::string tok;
            for (int c = 0; c < ncol && (ss >> tok); ++c)
                for (int d = 0; d < 3; ++d)
                    if (c == col[d]) p(d) = std::strtod(tok.c_str(), nullptr);
            if (keep(p)) pc.points_.push_back(p);
        }
    } else {
        return false;  // binary_compressed: not provided by the stand-in
    }
    return true;
}
inline bool ReadPointCloudFromPLY(const std::string &, geometry::PointCloud &, const ReadPointCloudOption &) {
    std::cerr << "stand-in Open3D: PLY is not provided" << std::endl;
    return false;
}
// binary PCD, xyz fp32 + packed rgb when the cloud has colours (the layout Open3D writes).
inline bool WritePointCloud(const std::string &path, const geometry::PointCloud &pc) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const bool rgb = pc.colors_.size() == pc.points_.size() && !pc.points_.empty();
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
      << (rgb ? "FIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n" : "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n")
      << "WIDTH " << pc.points_.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << pc.points_.size()
      << "\nDATA binary\n";
    for (size_t i = 0; i < pc.points_.size(); ++i) {
        float v[4] = {(float) pc.points_[i](0), (float) pc.points_[i](1), (float) pc.points_[i](2), 0.f};
        if (rgb) {
            auto u8 = [](double c) { return (std::uint32_t) std::round(std::min(1.0, std::max(0.0, c)) * 255.0); };
            const std::uint32_t packed = (u8(pc.colors_[i](0)) << 16) | (u8(pc.colors_[i](1)) << 8) | u8(pc.colors_[i](2));
            std::memcpy(&v[3], &packed, 4);
        }
        f.write((const char *) v, rgb ? 16 : 12);
    }
    return (bool) f;
}
inline bool WriteTriangleMesh(const std::string &, const geometry::TriangleMesh &) { return false; }

}  // namespace io

namespace visualization {

class ColorMapJet {
    static double Interpolate(double value, double y0, double x0, double y1, double x1) {
        if (value < x0) return y0;
        if (value > x1) return y1;
        return (value - x0) * (y1 - y0) / (x1 - x0) + y0;
    }
    static double JetBase(double value) {
        if (value <= -0.75) return 0.0;
        if (value <= -0.25) return Interpolate(value, 0.0, -0.75, 1.0, -0.25);
        if (value <= 0.25) return 1.0;
        if (value <= 0.75) return Interpolate(value, 1.0, 0.25, 0.0, 0.75);
        return 0.0;
    }

  public:
    Eigen::Vector3d GetColor(double value) const {
        return Eigen::Vector3d(JetBase(value * 2.0 - 1.5), JetBase(value * 2.0 - 1.0), JetBase(value * 2.0 - 0.5));
    }
};

inline bool DrawGeometries(const std::vector<std::shared_ptr<const geometry::Geometry>> &, const std::string & = "Open3D") {
    return true;  // no window in the stand-in
}

}  // namespace visualization

namespace pipelines {
namespace registration {

typedef std::vector<Eigen::Vector2i> CorrespondenceSet;

class RegistrationResult {
  public:
    RegistrationResult(const Eigen::Matrix4d &T = Eigen::Matrix4d::Identity()) : transformation_(T) {}
    Eigen::Matrix4d transformation_;
    CorrespondenceSet correspondence_set_;
    double inlier_rmse_ = 0.0;
    double fitness_ = 0.0;
};
class ICPConvergenceCriteria {
  public:
    ICPConvergenceCriteria(double rf = 1e-6, double rr = 1e-6, int it = 30)
        : relative_fitness_(rf), relative_rmse_(rr), max_iteration_(it) {}
    double relative_fitness_, relative_rmse_;
    int max_iteration_;
};
class TransformationEstimation {};
class TransformationEstimationPointToPoint : public TransformationEstimation {};
class TransformationEstimationPointToPlane : public TransformationEstimation {};
class TransformationEstimationForGeneralizedICP : public TransformationEstimation {};

// GetRegistrationResultAndCorrespondences: every source point whose nearest target point lies within max_distance
// (SearchHybrid(q, max, 1): d2 < max^2) gives the pair (source index, target index), in source order.
inline RegistrationResult EvaluateRegistration(const geometry::PointCloud &source, const geometry::PointCloud &target,
                                               double max_correspondence_distance,
                                               const Eigen::Matrix4d &transformation = Eigen::Matrix4d::Identity()) {
    RegistrationResult result(transformation);
    geometry::PointCloud pcd = source;
    if (!(transformation == Eigen::Matrix4d::Identity())) pcd.Transform(transformation);
    geometry::KDTreeFlann kdtree(target);
    double error2 = 0.0;
    std::vector<int> idx(1);
    std::vector<double> d2(1);
    for (size_t i = 0; i < pcd.points_.size(); ++i)
        if (kdtree.SearchKNN(pcd.points_[i], 1, idx, d2) > 0 && d2[0] < max_correspondence_distance * max_correspondence_distance) {
            error2 += d2[0];
            result.correspondence_set_.push_back(Eigen::Vector2i((int) i, idx[0]));
        }
    if (!result.correspondence_set_.empty()) {
        result.fitness_ = (double) result.correspondence_set_.size() / (double) pcd.points_.size();
        result.inlier_rmse_ = std::sqrt(error2 / (double) result.correspondence_set_.size());
    }
    return result;
}
inline RegistrationResult RegistrationICP(const geometry::PointCloud &, const geometry::PointCloud &, double,
                                          const Eigen::Matrix4d &, const TransformationEstimation &,
                                          const ICPConvergenceCriteria &) {
    throw std::runtime_error("stand-in Open3D: the ICP solvers are not provided");
}
inline RegistrationResult RegistrationGeneralizedICP(const geometry::PointCloud &, const geometry::PointCloud &, double,
                                                     const Eigen::Matrix4d &, const TransformationEstimationForGeneralizedICP &,
                                                     const ICPConvergenceCriteria &) {
    throw std::runtime_error("stand-in Open3D: the ICP solvers are not provided");
}

}  // namespace registration
}  // namespace pipelines

}  // namespace open3d
