// mapeval_oracle.cpp — CPU ORACLE (test infrastructure only; see mapeval_oracle.h for the pinning status).
//
// Restates, function by function, the metric hot path of the reference
// (all citations relative to /root/reference/map_eval/src/).  Built with -ffp-contract=off so that the
// squared distance ((dx*dx + dy*dy) + dz*dz) has one well-defined fp64 value that the HIP engine must match
// bit for bit (inlier counts depend on it).
#include "mapeval_oracle.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <unordered_map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int kLeafMax = 15;  // Open3D KDTreeFlann -> nanoflann KDTreeSingleIndexAdaptorParams(15) [upstream]

inline double dist2(const double *a, const double *b) {
    // nanoflann L2_Simple_Adaptor accumulate order for dim 3 [upstream]: ((dx^2 + dy^2) + dz^2)
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

// ---------------------------------------------------------------------------------------------
// KD-tree: sliding-midpoint split on the widest bbox dimension, leaves of <= 15 points, index
// indirection (points are not reordered) — the published nanoflann construction.
// ---------------------------------------------------------------------------------------------
struct Node {
    // leaf: left/right = [begin,end) into vind, child1 = -1
    int32_t child1 = -1, child2 = -1;
    int32_t left = 0, right = 0;
    int32_t divfeat = 0;
    double divlow = 0, divhigh = 0;
};

}  // namespace

struct orc_kdtree {
    const double *pts = nullptr;
    int64_t n = 0;
    std::vector<int32_t> vind;
    // Nodes live in one malloc'ed array of 2n records handed out through an atomic counter (a binary tree over n points has
    // at most 2n - 1 nodes; untouched pages are never committed), so that sub-trees can be built by concurrent OpenMP tasks.
    // The tree SHAPE does not depend on the thread count (every split looks at its own point range only); only the
    // numbering of the nodes does, and no result depends on that.
    Node *nodes = nullptr;
    std::atomic<int64_t> n_nodes{0};
    double bb_lo[3], bb_hi[3];
    int32_t root = -1;
    int64_t task_cutoff = 0;  // sub-trees above this size are split into OpenMP tasks (0 = serial build)
    ~orc_kdtree() { std::free(nodes); }

    int32_t build(int32_t left, int32_t right, double lo[3], double hi[3]) {
        const int32_t id = (int32_t) n_nodes.fetch_add(1, std::memory_order_relaxed);
        nodes[id] = Node();
        if (right - left <= kLeafMax) {
            nodes[id].left = left;
            nodes[id].right = right;
            for (int d = 0; d < 3; ++d) {
                double mn = pts[3 * (int64_t) vind[left] + d], mx = mn;
                for (int32_t k = left + 1; k < right; ++k) {
                    const double v = pts[3 * (int64_t) vind[k] + d];
                    mn = std::min(mn, v);
                    mx = std::max(mx, v);
                }
                lo[d] = mn;
                hi[d] = mx;
            }
            return id;
        }
        // widest bbox dimension; among near-widest dims pick the one with the largest point spread
        double max_span = hi[0] - lo[0];
        for (int d = 1; d < 3; ++d) max_span = std::max(max_span, hi[d] - lo[d]);
        int cut = 0;
        double max_spread = -1.0;
        for (int d = 0; d < 3; ++d) {
            if (hi[d] - lo[d] > (1.0 - 1e-5) * max_span) {
                double mn = pts[3 * (int64_t) vind[left] + d], mx = mn;
                for (int32_t k = left + 1; k < right; ++k) {
                    const double v = pts[3 * (int64_t) vind[k] + d];
                    mn = std::min(mn, v);
                    mx = std::max(mx, v);
                }
                if (mx - mn > max_spread) {
                    max_spread = mx - mn;
                    cut = d;
                }
            }
        }
        double mn = pts[3 * (int64_t) vind[left] + cut], mx = mn;
        for (int32_t k = left + 1; k < right; ++k) {
            const double v = pts[3 * (int64_t) vind[k] + cut];
            mn = std::min(mn, v);
            mx = std::max(mx, v);
        }
        double cutval = 0.5 * (lo[cut] + hi[cut]);
        cutval = std::min(std::max(cutval, mn), mx);
        // three-way partition: [< cutval | == cutval | > cutval]
        int32_t l = left, r = right - 1;
        for (;;) {
            while (l <= r && pts[3 * (int64_t) vind[l] + cut] < cutval) ++l;
            while (l <= r && pts[3 * (int64_t) vind[r] + cut] >= cutval) --r;
            if (l > r) break;
            std::swap(vind[l], vind[r]);
            ++l;
            --r;
        }
        const int32_t lim1 = l;
        r = right - 1;
        for (;;) {
            while (l <= r && pts[3 * (int64_t) vind[l] + cut] <= cutval) ++l;
            while (l <= r && pts[3 * (int64_t) vind[r] + cut] > cutval) --r;
            if (l > r) break;
            std::swap(vind[l], vind[r]);
            ++l;
            --r;
        }
        const int32_t lim2 = l;
        const int32_t half = (right - left) / 2;
        int32_t idx;
        if (lim1 - left > half) idx = lim1;
        else if (lim2 - left < half) idx = lim2;
        else idx = left + half;
        if (idx == left) idx = left + 1;  // cannot happen unless all points coincide on `cut`
        if (idx == right) idx = right - 1;

        double lo1[3] = {lo[0], lo[1], lo[2]}, hi1[3] = {hi[0], hi[1], hi[2]};
        double lo2[3] = {lo[0], lo[1], lo[2]}, hi2[3] = {hi[0], hi[1], hi[2]};
        hi1[cut] = cutval;
        lo2[cut] = cutval;
        int32_t c1, c2;
        if (task_cutoff > 0 && right - left > task_cutoff) {
#pragma omp task shared(c1, lo1, hi1) default(shared)
            c1 = build(left, idx, lo1, hi1);
#pragma omp task shared(c2, lo2, hi2) default(shared)
            c2 = build(idx, right, lo2, hi2);
#pragma omp taskwait
        } else {
            c1 = build(left, idx, lo1, hi1);
            c2 = build(idx, right, lo2, hi2);
        }
        nodes[id].child1 = c1;
        nodes[id].child2 = c2;
        nodes[id].divfeat = cut;
        nodes[id].divlow = hi1[cut];
        nodes[id].divhigh = lo2[cut];
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo1[d], lo2[d]);
            hi[d] = std::max(hi1[d], hi2[d]);
        }
        return id;
    }

    // A subtree whose lower bound exceeds `worst` by more than this relative slack cannot hold a point
    // whose *computed* d2 is below `worst` (the incremental bound carries a few ulps of rounding).
    static constexpr double kSlack = 1.0 + 1e-13;

    void nn_rec(int32_t id, const double *q, double mindist, double dists[3], int32_t &best_i,
                double &best_d) const {
        const Node &nd = nodes[id];
        if (nd.child1 < 0) {
            for (int32_t k = nd.left; k < nd.right; ++k) {
                const int32_t pi = vind[k];
                const double d = dist2(q, pts + 3 * (int64_t) pi);
                if (d < best_d || (d == best_d && pi < best_i)) {
                    best_d = d;
                    best_i = pi;
                }
            }
            return;
        }
        const int f = nd.divfeat;
        const double val = q[f];
        const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int32_t first, other;
        double cut;
        if (diff1 + diff2 < 0) {
            first = nd.child1;
            other = nd.child2;
            cut = diff2 * diff2;
        } else {
            first = nd.child2;
            other = nd.child1;
            cut = diff1 * diff1;
        }
        nn_rec(first, q, mindist, dists, best_i, best_d);
        const double saved = dists[f];
        const double md = mindist + cut - saved;
        dists[f] = cut;
        if (md <= best_d * kSlack) nn_rec(other, q, md, dists, best_i, best_d);
        dists[f] = saved;
    }

    template <class F>
    void radius_rec(int32_t id, const double *q, double r2, double mindist, double dists[3], F &&emit) const {
        const Node &nd = nodes[id];
        if (nd.child1 < 0) {
            for (int32_t k = nd.left; k < nd.right; ++k) {
                const int32_t pi = vind[k];
                const double d = dist2(q, pts + 3 * (int64_t) pi);
                if (d < r2) emit(pi, d);  // nanoflann RadiusResultSet::addPoint: dist < radius [upstream]
            }
            return;
        }
        const int f = nd.divfeat;
        const double val = q[f];
        const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int32_t first, other;
        double cut;
        if (diff1 + diff2 < 0) {
            first = nd.child1;
            other = nd.child2;
            cut = diff2 * diff2;
        } else {
            first = nd.child2;
            other = nd.child1;
            cut = diff1 * diff1;
        }
        radius_rec(first, q, r2, mindist, dists, emit);
        const double saved = dists[f];
        const double md = mindist + cut - saved;
        dists[f] = cut;
        if (md <= r2 * kSlack) radius_rec(other, q, r2, md, dists, emit);
        dists[f] = saved;
    }

    double init_dists(const double *q, double dists[3]) const {
        double s = 0;
        for (int d = 0; d < 3; ++d) {
            dists[d] = 0;
            if (q[d] < bb_lo[d]) dists[d] = (q[d] - bb_lo[d]) * (q[d] - bb_lo[d]);
            if (q[d] > bb_hi[d]) dists[d] = (q[d] - bb_hi[d]) * (q[d] - bb_hi[d]);
            s += dists[d];
        }
        return s;
    }

    void nn1(const double *q, int32_t &best_i, double &best_d) const {
        best_i = -1;
        best_d = std::numeric_limits<double>::infinity();
        if (n == 0) return;
        double dists[3];
        const double md = init_dists(q, dists);
        nn_rec(root, q, md, dists, best_i, best_d);
    }

    template <class F>
    void radius(const double *q, double r2, F &&emit) const {
        if (n == 0) return;
        double dists[3];
        const double md = init_dists(q, dists);
        radius_rec(root, q, r2, md, dists, emit);
    }

    // k nearest neighbours, ascending by (d2, index): nanoflann KNNResultSet keeps its k best sorted by distance
    // [upstream]; equal distances are ordered by point index here (the traversal order decides upstream).
    using Hit = std::pair<double, int32_t>;
    void knn_rec(int32_t id, const double *q, double mindist, double dists[3], size_t k, std::vector<Hit> &res) const {
        const Node &nd = nodes[id];
        if (nd.child1 < 0) {
            for (int32_t kk = nd.left; kk < nd.right; ++kk) {
                const int32_t pi = vind[kk];
                const Hit h(dist2(q, pts + 3 * (int64_t) pi), pi);
                if (res.size() < k || h < res.back()) {
                    res.insert(std::upper_bound(res.begin(), res.end(), h), h);
                    if (res.size() > k) res.pop_back();
                }
            }
            return;
        }
        const int f = nd.divfeat;
        const double val = q[f];
        const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int32_t first, other;
        double cut;
        if (diff1 + diff2 < 0) {
            first = nd.child1;
            other = nd.child2;
            cut = diff2 * diff2;
        } else {
            first = nd.child2;
            other = nd.child1;
            cut = diff1 * diff1;
        }
        knn_rec(first, q, mindist, dists, k, res);
        const double saved = dists[f];
        const double md = mindist + cut - saved;
        dists[f] = cut;
        const double worst = res.size() < k ? std::numeric_limits<double>::infinity() : res.back().first;
        if (md <= worst * kSlack) knn_rec(other, q, md, dists, k, res);
        dists[f] = saved;
    }
    void knn(const double *q, size_t k, std::vector<Hit> &res) const {
        res.clear();
        if (n == 0 || k == 0) return;
        double dists[3];
        const double md = init_dists(q, dists);
        knn_rec(root, q, md, dists, k, res);
    }
};

namespace {

// ---------------------------------------------------------------------------------------------
// small 3x3 linear algebra (Eigen restated)
// ---------------------------------------------------------------------------------------------
inline double det3(const double m[9]) {
    // Eigen determinant_impl<Derived,3>: m00*(m11*m22-m12*m21) - m01*(m10*m22-m12*m20) + m02*(m10*m21-m11*m20)
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (stands in for SelfAdjointEigenSolver<Matrix3d>;
// the clamped reconstruction V*max(L,1e-6)*V^T does not depend on the solver beyond rounding).
void jacobi_eig3(const double a_in[9], double evals[3], double V[9]) {
    double a[9];
    std::memcpy(a, a_in, sizeof(a));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p) {
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[3 * p + q];
                if (apq == 0.0) continue;
                const double app = a[3 * p + p], aqq = a[3 * q + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                // A <- J^T A J
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - s * akq;
                    a[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = c * apk - s * aqk;
                    a[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq;
                    V[3 * k + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    evals[0] = a[0];
    evals[1] = a[4];
    evals[2] = a[8];
}

// Lower Cholesky factor (Eigen LLT, no pivoting, info() unchecked as in voxel_calculator.cpp:136-137).
void chol3(const double a[9], double L[9]) {
    for (int i = 0; i < 9; ++i) L[i] = 0.0;
    L[0] = std::sqrt(a[0]);
    L[3] = a[3] / L[0];
    L[6] = a[6] / L[0];
    L[4] = std::sqrt(a[4] - L[3] * L[3]);
    L[7] = (a[7] - L[6] * L[3]) / L[4];
    L[8] = std::sqrt(a[8] - L[6] * L[6] - L[7] * L[7]);
}

// voxel_calculator.cpp:119-125 / :127-133
void regularize_sigma(const double sigma_stored[9], int n, double out[9]) {
    if (n > 1) {
        double s[9];
        for (int i = 0; i < 9; ++i) s[i] = sigma_stored[i] / (double) (n - 1);  // third division (:120)
        double sym[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) sym[3 * r + c] = (s[3 * r + c] + s[3 * c + r]) / 2;  // (:121)
        double ev[3], V[9];
        jacobi_eig3(sym, ev, V);
        for (int k = 0; k < 3; ++k) ev[k] = std::max(ev[k], 1e-6);  // cwiseMax(1e-6) (:123)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += V[3 * r + k] * ev[k] * V[3 * c + k];
                out[3 * r + c] = acc;
            }
    } else {
        for (int i = 0; i < 9; ++i) out[i] = (i % 4 == 0) ? 1.0 : 0.0;  // Identity (:118,:126)
    }
}

struct VoxelInfo {  // voxel_calculator.hpp:25-38
    double mu[3] = {0, 0, 0};
    double sigma[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int num_points = 0;
    double entropy = 0, energy = 0;
    int active = 0;
    double entropy_old = 0;
};

struct Key3 {
    int32_t v[3];
    bool operator==(const Key3 &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
struct KeyHash {  // voxel_calculator.hpp:18-22 / map_eval.h:53-58: XOR of std::hash<int>
    size_t operator()(const Key3 &k) const {
        return std::hash<int>()(k.v[0]) ^ std::hash<int>()(k.v[1]) ^ std::hash<int>()(k.v[2]);
    }
};
inline bool key_less(const Key3 &a, const Key3 &b) {
    if (a.v[0] != b.v[0]) return a.v[0] < b.v[0];
    if (a.v[1] != b.v[1]) return a.v[1] < b.v[1];
    return a.v[2] < b.v[2];
}

using VoxelMap = std::unordered_map<Key3, VoxelInfo, KeyHash>;

void compute_voxel_entropy(VoxelInfo &v) {  // voxel_calculator.cpp:97-113
    if (v.num_points < 2) {
        v.entropy = 0;
        v.energy = 0;
    } else {
        for (int i = 0; i < 9; ++i) v.sigma[i] /= (double) (v.num_points - 1);  // second division (:102)
        const double det = det3(v.sigma);
        if (det <= 0) {
            v.entropy = 0;
            v.energy = 0;
        } else {
            constexpr double PI = 3.141592653589793238463;
            v.entropy = 0.5 * std::log(std::pow(2 * PI * std::exp(1), 3) * det);
            v.energy = v.sigma[0] + v.sigma[4] + v.sigma[8];
        }
    }
}

double w2_gaussian(const VoxelInfo &v1, const VoxelInfo &v2) {  // voxel_calculator.cpp:115-140
    double s1[9], s2[9];
    regularize_sigma(v1.sigma, v1.num_points, s1);
    regularize_sigma(v2.sigma, v2.num_points, s2);
    double md = 0;
    for (int d = 0; d < 3; ++d) md += (v1.mu[d] - v2.mu[d]) * (v1.mu[d] - v2.mu[d]);
    const double tr_sum = (s1[0] + s2[0]) + (s1[4] + s2[4]) + (s1[8] + s2[8]);
    double L1[9];
    chol3(s1, L1);
    // M = L1 * sigma2 * L1^T  (:136)
    double T[9], M[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += L1[3 * r + k] * s2[3 * k + c];
            T[3 * r + c] = acc;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += T[3 * r + k] * L1[3 * c + k];
            M[3 * r + c] = acc;
        }
    double L[9];
    chol3(M, L);  // (:137)
    const double distance = md + tr_sum - 2 * (L[0] + L[4] + L[8]);  // (:138)
    return std::sqrt(std::max(0.0, distance));                        // (:139)
}

inline int32_t voxel_index(double x, double vs) { return (int32_t) std::floor(x / vs); }  // :241-245

std::vector<std::pair<Key3, const VoxelInfo *>> sorted_entries(const VoxelMap &m) {
    std::vector<std::pair<Key3, const VoxelInfo *>> v;
    v.reserve(m.size());
    for (const auto &kv : m) v.emplace_back(kv.first, &kv.second);
    std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return key_less(a.first, b.first); });
    return v;
}

double scs_from_map(const std::unordered_map<Key3, double, KeyHash> &wd, int radius) {  // map_eval.cpp:347-389
    double total_scs = 0.0;
    int64_t scs_count = 0;
    std::vector<double> nb;
    for (const auto &kv : wd) {
        nb.clear();
        for (int dx = -radius; dx <= radius; ++dx)       // getNeighborIndices, voxel_calculator.cpp:7-19
            for (int dy = -radius; dy <= radius; ++dy)
                for (int dz = -radius; dz <= radius; ++dz) {
                    if (dx == 0 && dy == 0 && dz == 0) continue;
                    Key3 k{{kv.first.v[0] + dx, kv.first.v[1] + dy, kv.first.v[2] + dz}};
                    auto it = wd.find(k);
                    if (it != wd.end()) nb.push_back(it->second);
                }
        if (!nb.empty()) {
            const double mean = std::accumulate(nb.begin(), nb.end(), 0.0) / nb.size();
            double var = 0.0;
            for (double w : nb) var += (w - mean) * (w - mean);
            var /= nb.size();
            total_scs += std::sqrt(var) / mean;
            scs_count++;
        }
    }
    return total_scs / (double) scs_count;  // NaN when scs_count == 0, as the reference (:387)
}

int resolve_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) return omp_get_max_threads();
    return threads;
#else
    (void) threads;
    return 1;
#endif
}

// one MME point (map_eval.cpp:1666-1701): returns true and sets H if the point is valid
bool mme_point(const orc_kdtree &t, int64_t i, double r2, int min_k, std::vector<std::pair<double, int32_t>> &nb,
               double &H) {
    const double *q = t.pts + 3 * i;
    nb.clear();
    t.radius(q, r2, [&](int32_t pi, double d) { nb.emplace_back(d, pi); });
    if (nb.empty()) return false;              // SearchRadius(...) > 0 (:1670)
    std::sort(nb.begin(), nb.end());           // nanoflann SearchParams(sorted = true) [upstream]
    nb.erase(nb.begin());                      // drop the query itself (:1672-1673)
    const size_t k = nb.size();
    if ((int) k < min_k) return false;         // (:1675 / :1458)
    double mean[3] = {0, 0, 0};
    for (size_t j = 0; j < k; ++j)
        for (int d = 0; d < 3; ++d) mean[d] += t.pts[3 * (int64_t) nb[j].second + d];
    for (int d = 0; d < 3; ++d) mean[d] /= (double) k;  // rowwise().mean() (:1684)
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t j = 0; j < k; ++j) {
        const double *p = t.pts + 3 * (int64_t) nb[j].second;
        const double c[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};  // (:1685)
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) cov[3 * r + cc] += c[r] * c[cc];
    }
    for (int e = 0; e < 9; ++e) cov[e] /= (double) (k - 1);                    // (:1688-1689)
    H = 0.5 * std::log(2 * M_PI * M_E * det3(cov));                             // (:1656)
    return !std::isnan(H) && !std::isinf(H);                                    // (:1692)
}

}  // namespace

struct orc_voxelmap {
    VoxelMap map;
    double voxel_size = 0;
    double total_entropy = 0, total_energy = 0;
};

extern "C" {

orc_kdtree *orc_kdtree_build_mt(const double *xyz, int64_t n, int threads) {
    // threads == 1: the reference's single-threaded SetGeometry (map_eval.cpp:1214 ...); otherwise the same tree built
    // by OpenMP tasks (test infrastructure for the 20 M / 50 M-point parity checks and the "all-parallel" CPU timing)
    auto *t = new orc_kdtree();
    t->pts = xyz;
    t->n = n;
    t->vind.resize((size_t) n);
    std::iota(t->vind.begin(), t->vind.end(), 0);
    if (n > 0) {
        for (int d = 0; d < 3; ++d) {
            double mn = xyz[d], mx = mn;
            for (int64_t i = 1; i < n; ++i) {
                mn = std::min(mn, xyz[3 * i + d]);
                mx = std::max(mx, xyz[3 * i + d]);
            }
            t->bb_lo[d] = mn;
            t->bb_hi[d] = mx;
        }
        t->nodes = static_cast<Node *>(std::malloc((size_t) (2 * n + 2) * sizeof(Node)));
        double lo[3] = {t->bb_lo[0], t->bb_lo[1], t->bb_lo[2]}, hi[3] = {t->bb_hi[0], t->bb_hi[1], t->bb_hi[2]};
        // (more than a few dozen threads only fight over the task queue and the first touch of the node array: on a
        // 256-core host the task build with all cores was slower than the serial one)
        const int nt = std::min(resolve_threads(threads), 32);
        if (nt > 1 && n > 100000) {
            t->task_cutoff = std::max<int64_t>(20000, n / (64 * (int64_t) nt));
#pragma omp parallel num_threads(nt)
#pragma omp single
            t->root = t->build(0, (int32_t) n, lo, hi);
        } else {
            t->root = t->build(0, (int32_t) n, lo, hi);
        }
    }
    return t;
}

orc_kdtree *orc_kdtree_build(const double *xyz, int64_t n) { return orc_kdtree_build_mt(xyz, n, 1); }

void orc_kdtree_free(orc_kdtree *t) { delete t; }

void orc_kdtree_nn1(const orc_kdtree *t, const double *q, int64_t m, int32_t *idx, double *d2, int threads) {
    const int nt = resolve_threads(threads);  // 1 = serial, 0 = all cores
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < m; ++i) {
        int32_t bi;
        double bd;
        t->nn1(q + 3 * i, bi, bd);
        if (idx) idx[i] = bi;
        if (d2) d2[i] = bd;
    }
}

void orc_kdtree_radius_count(const orc_kdtree *t, const double *q, int64_t m, double r, int32_t *count,
                             int threads) {
    const int nt = resolve_threads(threads);  // 1 = serial, 0 = all cores
    const double r2 = r * r;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < m; ++i) {
        int32_t c = 0;
        t->radius(q + 3 * i, r2, [&](int32_t, double) { ++c; });
        count[i] = c;
    }
}

void orc_transform(double *xyz, int64_t n, const double T[16]) {
    // Open3D PointCloud::Transform [upstream]: new = T * (x,y,z,1); p = new.head<3>() / new(3).
    // Column-major accumulation order of Eigen's 4x4 * 4x1 product: ((c0*x + c1*y) + c2*z) + c3*1.
    for (int64_t i = 0; i < n; ++i) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        double h[4];
        for (int r = 0; r < 4; ++r) h[r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
        xyz[3 * i] = h[0] / h[3];
        xyz[3 * i + 1] = h[1] / h[3];
        xyz[3 * i + 2] = h[2] / h[3];
    }
}

void orc_reg_stats_run(const double *src, int64_t ns, const double *tgt, int64_t nt_, double gate, int gate_mode,
                       const double trunc[5], orc_reg_stats *out, int threads) {
    orc_kdtree *tree = orc_kdtree_build(tgt, nt_);
    std::vector<double> d2((size_t) ns);
    std::vector<int32_t> idx((size_t) ns);
    orc_kdtree_nn1(tree, src, ns, idx.data(), d2.data(), threads);  // (:1215-1223), serial in the reference
    orc_kdtree_free(tree);

    std::memset(out, 0, sizeof(*out));
    out->n_src = ns;
    std::vector<double> dis;  // est_gt_dis (:1076)
    dis.reserve((size_t) ns);
    double number[5] = {0}, mean[5] = {0}, rmse[5] = {0};
    double sum_sqrt_all = 0;
    for (int64_t i = 0; i < ns; ++i) {
        if (idx[i] < 0) continue;  // SearchKNN(...) > 0
        sum_sqrt_all += std::sqrt(d2[i]);  // computeChamferDistance (:1416), ungated
        bool keep;
        if (gate < 0) keep = true;
        else if (gate_mode == 0) keep = d2[i] <= gate;          // (:1219) squared vs un-squared, sic
        else keep = d2[i] < gate * gate;                         // Open3D hybrid search [upstream]
        if (!keep) continue;
        const double norm_dis = std::sqrt(d2[i]);  // (map_pt - gt_pt).norm() (:1095)
        const double squre_dis = d2[i];            // squaredNorm (:1096)
        dis.push_back(norm_dis);
        for (int k = 0; k < 5; ++k)
            if (norm_dis <= trunc[k]) {  // (:1099-1123)
                mean[k] += norm_dis;
                rmse[k] += squre_dis;
                number[k] += 1.0;
            }
    }
    const double C = (double) dis.size();
    out->n_corr = (int64_t) dis.size();
    out->sum_sqrt_all = sum_sqrt_all;
    for (int k = 0; k < 5; ++k) {
        mean[k] /= C;  // (:1125)  NaN when C == 0, as the reference
        rmse[k] /= C;  // (:1126)
        out->number[k] = number[k];
        out->mean[k] = mean[k];
        out->fitness[k] = number[k] * 1.0 / (double) ns;  // (:1130) denominator = source.size()
        out->rmse[k] = std::sqrt(rmse[k]);                // (:1131)
        double sigma = 0.0;
        for (size_t j = 0; j < dis.size(); ++j) {
            const double e = dis[j] - mean[k];
            sigma += e * e;  // std::pow(error_dis, 2) (:1135)
        }
        sigma /= C;                       // (:1137)
        out->sigma[k] = std::sqrt(sigma);  // (:1138)
    }
}

double orc_chamfer(const double *a, int64_t na, const double *b, int64_t nb, int threads) {
    // map_eval.cpp:1398-1431
    orc_kdtree *ta = orc_kdtree_build(a, na);
    orc_kdtree *tb = orc_kdtree_build(b, nb);
    const int nt = resolve_threads(threads);
    double sum_p_to_q = 0.0, sum_q_to_p = 0.0;
#pragma omp parallel for reduction(+ : sum_p_to_q) num_threads(nt)
    for (int64_t i = 0; i < na; ++i) {
        int32_t bi;
        double bd;
        tb->nn1(a + 3 * i, bi, bd);
        if (bi >= 0) sum_p_to_q += std::sqrt(bd);
    }
#pragma omp parallel for reduction(+ : sum_q_to_p) num_threads(nt)
    for (int64_t i = 0; i < nb; ++i) {
        int32_t bi;
        double bd;
        ta->nn1(b + 3 * i, bi, bd);
        if (bi >= 0) sum_q_to_p += std::sqrt(bd);
    }
    orc_kdtree_free(ta);
    orc_kdtree_free(tb);
    return sum_p_to_q / (double) na + sum_q_to_p / (double) nb;  // (:1429)
}

double orc_mme(const double *xyz, int64_t n, double radius, int min_k, double *entropies, uint8_t *valid,
               int64_t *n_valid, double *sum_entropy, int mode, int threads) {
    orc_kdtree *tree = orc_kdtree_build(xyz, n);  // (:1618-1619)
    const double r2 = radius * radius;            // Open3D SearchRadius -> radiusSearch(q, r*r) [upstream]
    if (entropies) std::fill(entropies, entropies + n, 0.0);  // (:1614)
    if (valid) std::fill(valid, valid + n, (uint8_t) 0);      // (:1615)
    double sum = 0.0;
    int64_t count = 0;
    const int nt = (mode == 0) ? 1 : resolve_threads(threads);
    if (mode == 0 || nt == 1) {
        std::vector<std::pair<double, int32_t>> nb;
        for (int64_t i = 0; i < n; ++i) {
            double H;
            if (mme_point(*tree, i, r2, min_k, nb, H)) {
                sum += H;
                if (entropies) entropies[i] = H;
                if (valid) valid[i] = 1;
                ++count;
            }
        }
    } else if (mode == 1) {
        // ComputeMeanMapEntropyUsingNormal (:1553): #pragma omp parallel for reduction(+)
#pragma omp parallel num_threads(nt) reduction(+ : sum, count)
        {
            std::vector<std::pair<double, int32_t>> nb;
#pragma omp for
            for (int64_t i = 0; i < n; ++i) {
                double H;
                if (mme_point(*tree, i, r2, min_k, nb, H)) {
                    sum += H;
                    if (entropies) entropies[i] = H;
                    if (valid) valid[i] = 1;
                    ++count;
                }
            }
        }
    } else {
        // tbb::parallel_reduce over blocked_range(0, N, grain), grain = max(1, N/(8*hw_threads)) (:1716-1717):
        // independent per-range partial sums joined afterwards (:1704-1708).
        const int64_t grain = std::max<int64_t>(1, n / (8 * (int64_t) nt));
        const int64_t nblocks = (n + grain - 1) / grain;
        std::vector<double> psum((size_t) nblocks, 0.0);
        std::vector<int64_t> pcnt((size_t) nblocks, 0);
#pragma omp parallel num_threads(nt)
        {
            std::vector<std::pair<double, int32_t>> nb;
            nb.reserve(100);  // (:1663)
#pragma omp for schedule(dynamic, 1)
            for (int64_t b = 0; b < nblocks; ++b) {
                const int64_t lo = b * grain, hi = std::min(n, lo + grain);
                double s = 0.0;
                int64_t c = 0;
                for (int64_t i = lo; i < hi; ++i) {
                    double H;
                    if (mme_point(*tree, i, r2, min_k, nb, H)) {
                        s += H;
                        if (entropies) entropies[i] = H;
                        if (valid) valid[i] = 1;
                        ++c;
                    }
                }
                psum[(size_t) b] = s;
                pcnt[(size_t) b] = c;
            }
        }
        for (int64_t b = 0; b < nblocks; ++b) {
            sum += psum[(size_t) b];
            count += pcnt[(size_t) b];
        }
    }
    orc_kdtree_free(tree);
    if (n_valid) *n_valid = count;
    if (sum_entropy) *sum_entropy = sum;
    return count > 0 ? sum / (double) count : 0.0;  // (:1720-1724)
}

void orc_mme_points(const orc_kdtree *tree, const int64_t *sel, int64_t m, double radius, int min_k, double *entropies,
                    uint8_t *valid, int threads) {
    // the per-point body of ComputeMeanMapEntropy* (map_eval.cpp:1666-1701) for the points sel[0..m) of the tree's own
    // cloud, against the FULL tree: entropies[j] (0.0 where invalid) and valid[j] for j < m
    const double r2 = radius * radius;
    const int nt = resolve_threads(threads);
#pragma omp parallel num_threads(nt)
    {
        std::vector<std::pair<double, int32_t>> nb;
        nb.reserve(100);
#pragma omp for schedule(dynamic, 256)
        for (int64_t j = 0; j < m; ++j) {
            double H = 0.0;
            const bool ok = mme_point(*tree, sel[j], r2, min_k, nb, H);
            entropies[j] = ok ? H : 0.0;
            valid[j] = ok ? 1 : 0;
        }
    }
}

orc_voxelmap *orc_voxel_build(const double *xyz, int64_t n, double voxel_size) {
    // VoxelCalculator::buildVoxelMap(open3d) voxel_calculator.cpp:21-56
    auto *vm = new orc_voxelmap();
    vm->voxel_size = voxel_size;
    VoxelMap &map = vm->map;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = xyz + 3 * i;
        Key3 key{{voxel_index(p[0], voxel_size), voxel_index(p[1], voxel_size), voxel_index(p[2], voxel_size)}};
        auto it = map.find(key);
        if (it == map.end()) {
            VoxelInfo v;
            v.num_points = 1;
            v.mu[0] = p[0];
            v.mu[1] = p[1];
            v.mu[2] = p[2];
            v.active = 1;
            map.emplace(key, v);
        } else {
            VoxelInfo &v = it->second;
            v.num_points++;
            const double delta[3] = {p[0] - v.mu[0], p[1] - v.mu[1], p[2] - v.mu[2]};  // (:39)
            for (int d = 0; d < 3; ++d) v.mu[d] += delta[d] / v.num_points;             // (:40)
            const double d2[3] = {p[0] - v.mu[0], p[1] - v.mu[1], p[2] - v.mu[2]};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) v.sigma[3 * r + c] += delta[r] * d2[c];     // (:41)
            v.energy = v.sigma[0] + v.sigma[4] + v.sigma[8];
        }
    }
    for (auto &kv : map) {
        VoxelInfo &v = kv.second;
        if (v.num_points > 10) {                                                         // (:47)
            for (int e = 0; e < 9; ++e) v.sigma[e] /= (double) (v.num_points - 1);       // first division (:48)
            compute_voxel_entropy(v);                                                    // second division
            v.entropy_old = v.entropy;
            vm->total_entropy += v.entropy;
            vm->total_energy += v.energy;
        }
    }
    return vm;
}

void orc_voxel_free(orc_voxelmap *m) { delete m; }
int64_t orc_voxel_count(const orc_voxelmap *m) { return (int64_t) m->map.size(); }

void orc_voxel_export(const orc_voxelmap *m, int32_t *keys, int32_t *npts, double *mu, double *sigma,
                      double *entropy) {
    const auto ent = sorted_entries(m->map);
    for (size_t i = 0; i < ent.size(); ++i) {
        const VoxelInfo &v = *ent[i].second;
        if (keys)
            for (int d = 0; d < 3; ++d) keys[3 * i + d] = ent[i].first.v[d];
        if (npts) npts[i] = v.num_points;
        if (mu)
            for (int d = 0; d < 3; ++d) mu[3 * i + d] = v.mu[d];
        if (sigma)
            for (int e = 0; e < 9; ++e) sigma[9 * i + e] = v.sigma[e];
        if (entropy) entropy[i] = v.entropy;
    }
}

double orc_w2_gaussian(const double mu1[3], const double sigma1[9], int n1, const double mu2[3],
                       const double sigma2[9], int n2) {
    VoxelInfo a, b;
    std::memcpy(a.mu, mu1, sizeof(a.mu));
    std::memcpy(a.sigma, sigma1, sizeof(a.sigma));
    a.num_points = n1;
    std::memcpy(b.mu, mu2, sizeof(b.mu));
    std::memcpy(b.sigma, sigma2, sizeof(b.sigma));
    b.num_points = n2;
    return w2_gaussian(a, b);
}

int orc_awd_scs(const orc_voxelmap *gt, const orc_voxelmap *est, double voxel_size, int min_pts, int scs_radius,
                double *rows, double *w_sorted, int64_t *n_rows, double *awd, double *scs, int64_t counts[3]) {
    // updateVoxelMap(gt_map) voxel_calculator.cpp:142-172 — labels only (the est map itself is not mutated here;
    // inserted empty "old" voxels never reach the W loop because they have num_points 0 < 100).
    int64_t active = 0, old_area = 0, new_area = 0;
    for (const auto &kv : gt->map) {
        if (est->map.find(kv.first) != est->map.end()) active++;
        else old_area++;
    }
    new_area = (int64_t) est->map.size() - active;
    if (counts) {
        counts[0] = active;
        counts[1] = old_area;
        counts[2] = new_area;
    }
    std::unordered_map<Key3, double, KeyHash> wd;  // wasserstein_distances (map_eval.cpp:266)
    const auto ent = sorted_entries(est->map);
    const int64_t cap = n_rows ? *n_rows : 0;
    int64_t nr = 0;
    std::vector<double> ws;
    for (const auto &e : ent) {
        const VoxelInfo &ev = *e.second;
        auto git = gt->map.find(e.first);  // active == 1 <=> key in gt map (:274-276)
        if (git == gt->map.end()) continue;
        const VoxelInfo &gv = git->second;
        if (ev.num_points < min_pts || gv.num_points < min_pts) continue;  // (:280)
        const double w = w2_gaussian(gv, ev);                              // (:284) (gt, est) order
        wd[e.first] = w;
        ws.push_back(w);
        if (rows && nr < cap) {
            double *r = rows + 27 * nr;  // column order of voxel_errors.txt (:292-302)
            for (int d = 0; d < 3; ++d) r[d] = (double) e.first.v[d] * voxel_size;
            for (int d = 0; d < 3; ++d) r[3 + d] = ((double) e.first.v[d] + 1.0) * voxel_size;
            for (int d = 0; d < 3; ++d) r[6 + d] = ev.mu[d];
            r[9] = w;
            r[10] = gv.num_points;
            r[11] = ev.num_points;
            r[12] = ev.sigma[0]; r[13] = ev.sigma[1]; r[14] = ev.sigma[2];
            r[15] = ev.sigma[4]; r[16] = ev.sigma[5]; r[17] = ev.sigma[8];
            for (int d = 0; d < 3; ++d) r[18 + d] = gv.mu[d];
            r[21] = gv.sigma[0]; r[22] = gv.sigma[1]; r[23] = gv.sigma[2];
            r[24] = gv.sigma[4]; r[25] = gv.sigma[5]; r[26] = gv.sigma[8];
        }
        ++nr;
    }
    if (n_rows) *n_rows = nr;
    const double mean_ws = std::accumulate(ws.begin(), ws.end(), 0.0) / (double) ws.size();  // (:324) NaN if empty
    if (awd) *awd = mean_ws;
    if (w_sorted) {
        std::sort(ws.begin(), ws.end());  // (:330)
        for (int64_t i = 0; i < std::min<int64_t>(cap, (int64_t) ws.size()); ++i) w_sorted[i] = ws[(size_t) i];
    }
    if (scs) *scs = scs_from_map(wd, scs_radius);
    return 0;
}

double orc_scs(const int32_t *keys, const double *w, int64_t n, int radius) {
    std::unordered_map<Key3, double, KeyHash> wd;
    for (int64_t i = 0; i < n; ++i) wd[Key3{{keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]}}] = w[i];
    return scs_from_map(wd, radius);
}

int64_t orc_voxel_downsample(const double *xyz, int64_t n, double voxel_size, double *out, int64_t capacity) {
    // open3d::geometry::PointCloud::VoxelDownSample [upstream] (called at map_eval.cpp:38-39):
    //   voxel_min_bound = GetMinBound() - voxel_size * 0.5
    //   voxel_index     = floor((p - voxel_min_bound) / voxel_size)   (per component, int)
    //   AccumulatedPoint: point_ += p (cloud order), output = point_ / num_of_points_
    // Output order here: ascending (ix, iy, iz); Open3D's is its hash map's iteration order.
    if (n <= 0 || !(voxel_size > 0)) return 0;
    double mn[3] = {xyz[0], xyz[1], xyz[2]};
    for (int64_t i = 1; i < n; ++i)
        for (int d = 0; d < 3; ++d) mn[d] = std::min(mn[d], xyz[3 * i + d]);
    for (int d = 0; d < 3; ++d) mn[d] -= voxel_size * 0.5;
    struct Acc {
        double s[3] = {0, 0, 0};
        int64_t c = 0;
    };
    std::unordered_map<Key3, Acc, KeyHash> acc;
    for (int64_t i = 0; i < n; ++i) {
        Key3 k;
        for (int d = 0; d < 3; ++d) k.v[d] = (int32_t) std::floor((xyz[3 * i + d] - mn[d]) / voxel_size);
        Acc &a = acc[k];
        for (int d = 0; d < 3; ++d) a.s[d] += xyz[3 * i + d];
        a.c++;
    }
    std::vector<std::pair<Key3, const Acc *>> v;
    v.reserve(acc.size());
    for (const auto &kv : acc) v.emplace_back(kv.first, &kv.second);
    std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return key_less(a.first, b.first); });
    if (out)
        for (size_t i = 0; i < v.size() && (int64_t) i < capacity; ++i)
            for (int d = 0; d < 3; ++d) out[3 * i + d] = v[i].second->s[d] / (double) v[i].second->c;
    return (int64_t) v.size();
}

// ---- renderers (map_eval.cpp:586-607, :686-735) --------------------------------------------------------------
// open3d::visualization::ColorMapJet::GetColor [Open3D ColorMap.h/.cpp, upstream — not in the reference tree]:
//   (JetBase(2v - 1.5), JetBase(2v - 1.0), JetBase(2v - 0.5)), JetBase piecewise linear with
//   ColorMap::Interpolate(value, y0, x0, y1, x1) = (value - x0) * (y1 - y0) / (x1 - x0) + y0, clamped to [y0, y1] outside.
static double jet_interpolate(double value, double y0, double x0, double y1, double x1) {
    if (value < x0) return y0;
    if (value > x1) return y1;
    return (value - x0) * (y1 - y0) / (x1 - x0) + y0;
}
static double jet_base(double value) {
    if (value <= -0.75) return 0.0;
    if (value <= -0.25) return jet_interpolate(value, 0.0, -0.75, 1.0, -0.25);
    if (value <= 0.25) return 1.0;
    if (value <= 0.75) return jet_interpolate(value, 1.0, 0.25, 0.0, 0.75);
    return 0.0;
}
void orc_jet_color(double value, double rgb[3]) {
    rgb[0] = jet_base(value * 2.0 - 1.5);
    rgb[1] = jet_base(value * 2.0 - 1.0);
    rgb[2] = jet_base(value * 2.0 - 0.5);
}

// renderDistanceOnPointCloud (:586-607): eval_dis = SQUARED NN distance (SearchKNN returns d2, :579-580), clamped to
// `dis` (the unsquared truncation distance, sic), a = eval_dis / dis, colour = Jet(a).  d2[n] -> rgb[n][3].
void orc_render_distance(const double *d2, int64_t n, double dis, double *rgb) {
    for (int64_t i = 0; i < n; ++i) {
        double e = d2[i];
        if (e > dis) e = dis;  // (:591-595)
        orc_jet_color(e / dis, rgb + 3 * i);  // (:601-603)
    }
}

// ColorPointCloudByMME(pointcloud, entropies) (:686-735): range over the non-zero entropies (max_abs = |min|, min_abs =
// |max|, :696-699), then for the VALID points only, in cloud order: normalise |H|, log-map with epsilon 0.1, Jet.
// xyz_out / rgb_out: capacity rows of 3 (may be NULL to count).  Returns the number of valid points.
int64_t orc_render_entropy(const double *xyz, const double *entropies, const uint8_t *valid, int64_t n, double *xyz_out,
                           double *rgb_out, int64_t capacity, double *min_abs_out, double *max_abs_out) {
    double mn = INFINITY, mx = -INFINITY;
    for (int64_t i = 0; i < n; ++i)
        if (entropies[i] != 0.0) {
            mn = std::min(mn, entropies[i]);
            mx = std::max(mx, entropies[i]);
        }
    const double max_abs = std::fabs(mn), min_abs = std::fabs(mx);
    if (min_abs_out) *min_abs_out = min_abs;
    if (max_abs_out) *max_abs_out = max_abs;
    const double epsilon = 1e-1;
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!valid[i]) continue;
        if (xyz_out && rgb_out && m < capacity) {
            double ne = (std::fabs(entropies[i]) - min_abs) / (max_abs - min_abs);      // (:714-715)
            const double mapped = std::log(ne + epsilon);                               // (:719)
            ne = (mapped - std::log(epsilon)) / (std::log(1.0 + epsilon) - std::log(epsilon));  // (:721)
            orc_jet_color(ne, rgb_out + 3 * m);
            for (int d = 0; d < 3; ++d) xyz_out[3 * m + d] = xyz[3 * i + d];
        }
        ++m;
    }
    return m;
}


}  // extern "C"

// =============================================================================================================
// Registration (performICPRegistration, map_eval.cpp:1366-1394): the Open3D pieces behind registration_methods 1 / 2.
// None of Open3D is in the reference tree; what follows restates its published algorithms [upstream, stated 0.15.1]:
//   PointCloud::EstimateNormals(KDTreeSearchParamKNN(k))  -> EstimatePerPointCovariances + ComputeNormal(fast = true)
//   utility::ComputeCovariance (one-pass raw moments), FastEigen3x3 (Eberly, "A Robust Eigensolver for 3x3 Symmetric
//   Matrices"), InitializePointCloudForGeneralizedICP(epsilon = 1e-3), GetRotationFromE1ToX,
//   TransformationEstimationForGeneralizedICP / PointToPlane ::ComputeTransformation (the J^T J, J^T r sums).
// 3x3 products are accumulated as (a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j.
// =============================================================================================================
namespace {

using Hit = orc_kdtree::Hit;

inline void mat3_mul(const double a[9], const double b[9], double out[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
inline void mat3_mul_bt(const double a[9], const double b[9], double out[9]) {  // a * b^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = (a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1]) + a[3 * i + 2] * b[3 * j + 2];
}
inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
inline double dot3(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// Eigen Matrix3d::inverse(): cofactors, determinant expanded along column 0
inline void inv3(const double m[9], double out[9]) {
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
    const double invdet = 1.0 / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = cof(j, i) * invdet;
}

// utility::ComputeCovariance(points, indices): raw first and second moments, divided by the count
void o3d_covariance(const double *pts, const Hit *nb, size_t m, double cov[9]) {
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t j = 0; j < m; ++j) {
        const double *p = pts + 3 * (int64_t) nb[j].second;
        c[0] += p[0];
        c[1] += p[1];
        c[2] += p[2];
        c[3] += p[0] * p[0];
        c[4] += p[0] * p[1];
        c[5] += p[0] * p[2];
        c[6] += p[1] * p[1];
        c[7] += p[1] * p[2];
        c[8] += p[2] * p[2];
    }
    for (int e = 0; e < 9; ++e) c[e] /= (double) m;
    cov[0] = c[3] - c[0] * c[0];
    cov[4] = c[6] - c[1] * c[1];
    cov[8] = c[8] - c[2] * c[2];
    cov[1] = cov[3] = c[4] - c[0] * c[1];
    cov[2] = cov[6] = c[5] - c[0] * c[2];
    cov[5] = cov[7] = c[7] - c[1] * c[2];
}

void o3d_eigvec0(const double A[9], double eval0, double out[3]) {
    const double row0[3] = {A[0] - eval0, A[1], A[2]};
    const double row1[3] = {A[1], A[4] - eval0, A[5]};
    const double row2[3] = {A[2], A[5], A[8] - eval0};
    double r0xr1[3], r0xr2[3], r1xr2[3];
    cross3(row0, row1, r0xr1);
    cross3(row0, row2, r0xr2);
    cross3(row1, row2, r1xr2);
    const double d0 = dot3(r0xr1, r0xr1), d1 = dot3(r0xr2, r0xr2), d2 = dot3(r1xr2, r1xr2);
    double dmax = d0;
    int imax = 0;
    if (d1 > dmax) {
        dmax = d1;
        imax = 1;
    }
    if (d2 > dmax) imax = 2;
    const double *v = imax == 0 ? r0xr1 : (imax == 1 ? r0xr2 : r1xr2);
    const double len = std::sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
    for (int k = 0; k < 3; ++k) out[k] = v[k] / len;
}

void o3d_eigvec1(const double A[9], const double e0[3], double eval1, double out[3]) {
    double U[3], V[3];
    if (std::fabs(e0[0]) > std::fabs(e0[1])) {
        const double inv_length = 1.0 / std::sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv_length;
        U[1] = 0;
        U[2] = e0[0] * inv_length;
    } else {
        const double inv_length = 1.0 / std::sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0;
        U[1] = e0[2] * inv_length;
        U[2] = -e0[1] * inv_length;
    }
    cross3(e0, U, V);
    const double AU[3] = {(A[0] * U[0] + A[1] * U[1]) + A[2] * U[2], (A[1] * U[0] + A[4] * U[1]) + A[5] * U[2],
                          (A[2] * U[0] + A[5] * U[1]) + A[8] * U[2]};
    const double AV[3] = {(A[0] * V[0] + A[1] * V[1]) + A[2] * V[2], (A[1] * V[0] + A[4] * V[1]) + A[5] * V[2],
                          (A[2] * V[0] + A[5] * V[1]) + A[8] * V[2]};
    double m00 = dot3(U, AU) - eval1, m01 = dot3(U, AV), m11 = dot3(V, AV) - eval1;
    const double a00 = std::fabs(m00), a01 = std::fabs(m01), a11 = std::fabs(m11);
    if (a00 >= a11) {
        if (std::max(a00, a01) > 0) {
            if (a00 >= a01) {
                m01 /= m00;
                m00 = 1 / std::sqrt(1 + m01 * m01);
                m01 *= m00;
            } else {
                m00 /= m01;
                m01 = 1 / std::sqrt(1 + m00 * m00);
                m00 *= m01;
            }
            for (int k = 0; k < 3; ++k) out[k] = m01 * U[k] - m00 * V[k];
        } else {
            for (int k = 0; k < 3; ++k) out[k] = U[k];
        }
    } else {
        if (std::max(a11, a01) > 0) {
            if (a11 >= a01) {
                m01 /= m11;
                m11 = 1 / std::sqrt(1 + m01 * m01);
                m01 *= m11;
            } else {
                m11 /= m01;
                m01 = 1 / std::sqrt(1 + m11 * m11);
                m11 *= m01;
            }
            for (int k = 0; k < 3; ++k) out[k] = m11 * U[k] - m01 * V[k];
        } else {
            for (int k = 0; k < 3; ++k) out[k] = U[k];
        }
    }
}

// FastEigen3x3: the eigenvector of the smallest eigenvalue of a symmetric 3x3 (closed form, matrix pre-scaled by its
// largest coefficient)
void o3d_fast_eigen3x3(const double cov[9], double out[3]) {
    double A[9];
    double max_coeff = cov[0];
    for (int e = 1; e < 9; ++e) max_coeff = std::max(max_coeff, cov[e]);
    if (max_coeff == 0) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
    for (int e = 0; e < 9; ++e) A[e] = cov[e] / max_coeff;
    const double norm = (A[1] * A[1] + A[2] * A[2]) + A[5] * A[5];
    if (norm > 0) {
        const double q = ((A[0] + A[4]) + A[8]) / 3;
        const double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
        const double p = std::sqrt((((b00 * b00 + b11 * b11) + b22 * b22) + norm * 2) / 6);
        const double c00 = b11 * b22 - A[5] * A[5];
        const double c01 = A[1] * b22 - A[5] * A[2];
        const double c02 = A[1] * A[5] - b11 * A[2];
        const double det = ((b00 * c00 - A[1] * c01) + A[2] * c02) / ((p * p) * p);
        double half_det = det * 0.5;
        half_det = std::min(std::max(half_det, -1.0), 1.0);
        const double angle = std::acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        const double beta2 = std::cos(angle) * 2;
        const double beta0 = std::cos(angle + two_thirds_pi) * 2;
        const double beta1 = -(beta0 + beta2);
        const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
        double e0[3], e1[3], e2[3];
        if (half_det >= 0) {
            o3d_eigvec0(A, ev2, e2);
            if (ev2 < ev0 && ev2 < ev1) {
                for (int k = 0; k < 3; ++k) out[k] = e2[k];
                return;
            }
            o3d_eigvec1(A, e2, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) {
                for (int k = 0; k < 3; ++k) out[k] = e1[k];
                return;
            }
            cross3(e1, e2, out);
        } else {
            o3d_eigvec0(A, ev0, e0);
            if (ev0 < ev1 && ev0 < ev2) {
                for (int k = 0; k < 3; ++k) out[k] = e0[k];
                return;
            }
            o3d_eigvec1(A, e0, ev1, e1);
            if (ev1 < ev0 && ev1 < ev2) {
                for (int k = 0; k < 3; ++k) out[k] = e1[k];
                return;
            }
            cross3(e0, e1, out);
        }
    } else {  // diagonal matrix (the scaling is undone upstream before this test; the comparisons are scale-free)
        out[0] = out[1] = out[2] = 0;
        if (cov[0] < cov[4] && cov[0] < cov[8]) out[0] = 1;
        else if (cov[4] < cov[0] && cov[4] < cov[8]) out[1] = 1;
        else out[2] = 1;
    }
}

// GetRotationFromE1ToX (GeneralizedICP.cpp): Rodrigues rotation taking e1 = (1,0,0) onto the unit vector x
void o3d_rotation_e1_to_x(const double x[3], double R[9]) {
    const double v[3] = {0.0, -x[2], x[1]};  // e1.cross(x)
    const double c = x[0];                   // e1.dot(x)
    for (int e = 0; e < 9; ++e) R[e] = (e % 4 == 0) ? 1.0 : 0.0;
    if (c < -0.99) return;                   // (sic) near-opposite: identity
    const double sv[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    double sv2[9];
    mat3_mul(sv, sv, sv2);
    const double factor = 1 / (1 + c);
    for (int e = 0; e < 9; ++e) R[e] = (R[e] + sv[e]) + sv2[e] * factor;
}

}  // namespace

extern "C" {

void orc_kdtree_knn(const orc_kdtree *t, const double *q, int64_t m, int k, int32_t *idx, double *d2, int threads) {
    const int nt = resolve_threads(threads);
#pragma omp parallel num_threads(nt) if (nt > 1)
    {
        std::vector<Hit> res;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < m; ++i) {
            t->knn(q + 3 * i, (size_t) k, res);
            for (int j = 0; j < k; ++j) {
                const bool have = j < (int) res.size();
                if (idx) idx[i * k + j] = have ? res[(size_t) j].second : -1;
                if (d2) d2[i * k + j] = have ? res[(size_t) j].first : std::numeric_limits<double>::infinity();
            }
        }
    }
}

// PointCloud::EstimateNormals(KDTreeSearchParamKNN(knn)) on a cloud WITHOUT normals: SearchKNN(point, knn) (the point
// itself is its first neighbour); >= 3 neighbours -> ComputeCovariance, else identity; normal = FastEigen3x3, a zero
// vector becomes (0,0,1); no orientation step (there is no previous normal to agree with).
void orc_estimate_normals_knn(const double *xyz, int64_t n, int knn, double *normals, int threads) {
    orc_kdtree *t = orc_kdtree_build(xyz, n);
    const int nt = resolve_threads(threads);
#pragma omp parallel num_threads(nt) if (nt > 1)
    {
        std::vector<Hit> res;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            t->knn(xyz + 3 * i, (size_t) knn, res);
            double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (res.size() >= 3) o3d_covariance(xyz, res.data(), res.size(), cov);
            double nv[3];
            o3d_fast_eigen3x3(cov, nv);
            if (std::sqrt(dot3(nv, nv)) == 0.0) {
                nv[0] = nv[1] = 0;
                nv[2] = 1;
            }
            for (int d = 0; d < 3; ++d) normals[3 * i + d] = nv[d];
        }
    }
    orc_kdtree_free(t);
}

// InitializePointCloudForGeneralizedICP: covariance = Rx * diag(epsilon, 1, 1) * Rx^T with Rx = rotation e1 -> normal
void orc_gicp_covariances(const double *normals, int64_t n, double epsilon, double *cov) {
    const double Cd[9] = {epsilon, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int64_t i = 0; i < n; ++i) {
        double R[9], RC[9];
        o3d_rotation_e1_to_x(normals + 3 * i, R);
        mat3_mul(R, Cd, RC);
        mat3_mul_bt(RC, R, cov + 9 * i);
    }
}

// PointCloud::Transform on per-point attributes: normals n <- R n, covariances C <- R C R^T (R = T.block<3,3>(0,0))
void orc_rotate_attributes(double *normals, double *cov, int64_t n, const double T[16]) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    for (int64_t i = 0; i < n; ++i) {
        if (normals) {
            const double v[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
            for (int r = 0; r < 3; ++r) normals[3 * i + r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1]) + R[3 * r + 2] * v[2];
        }
        if (cov) {
            double RC[9], out[9];
            mat3_mul(R, cov + 9 * i, RC);
            mat3_mul_bt(RC, R, out);
            std::memcpy(cov + 9 * i, out, sizeof(out));
        }
    }
}

// One linearised least-squares step of RegistrationICP / RegistrationGeneralizedICP: correspondences = 1-NN of every
// source point in the target with d2 < max^2 (GetRegistrationResultAndCorrespondences), then the sums of
// utility::ComputeJTJandJTr over them.
//   mode 1, point-to-plane: J = [vs x nt, nt] (1 x 6), r = (vs - vt) . nt          (tgt_attr = target normals, n x 3)
//   mode 2, generalized   : M = Ct + Cs, W = M^(-1/2); rows of W [-skew(vs) | I], r = W (vs - vt);
//           sum_rows J^T J = [-skew(vs) | I]^T M^-1 [-skew(vs) | I] (W symmetric, W W = M^-1): the matrix square
//           root is not formed.                                                     (attrs = covariances, n x 9)
void orc_icp_lsq_sums(int mode, const double *src, const double *src_attr, int64_t ns, const double *tgt,
                      const double *tgt_attr, int64_t nt_, double max_distance, orc_lsq_sums *out, int threads) {
    std::vector<int32_t> idx((size_t) ns);
    std::vector<double> d2((size_t) ns);
    orc_kdtree *t = orc_kdtree_build(tgt, nt_);
    orc_kdtree_nn1(t, src, ns, idx.data(), d2.data(), threads);
    orc_kdtree_free(t);
    std::memset(out, 0, sizeof(*out));
    out->n_src = ns;
    const double gate2 = max_distance * max_distance;
    for (int64_t i = 0; i < ns; ++i) {
        if (idx[(size_t) i] < 0 || !(d2[(size_t) i] < gate2)) continue;
        const int64_t j = idx[(size_t) i];
        const double *vs = src + 3 * i, *vt = tgt + 3 * j;
        const double d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
        ++out->n_corr;
        out->sum_d2 += d2[(size_t) i];
        if (mode == 1) {
            const double *nt = tgt_attr + 3 * j;
            double J[6];
            cross3(vs, nt, J);
            J[3] = nt[0];
            J[4] = nt[1];
            J[5] = nt[2];
            const double r = dot3(d, nt);
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) out->JTJ[6 * a + b] += J[a] * J[b];
                out->JTr[a] += J[a] * r;
            }
            out->r2 += r * r;
        } else {
            double M[9], B[9];
            for (int e = 0; e < 9; ++e) M[e] = tgt_attr[9 * j + e] + src_attr[9 * i + e];
            inv3(M, B);
            const double x = vs[0], y = vs[1], z = vs[2];
            const double J[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};  // [-skew(vs) | I], 3 x 6
            double BJ[18];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 6; ++c) BJ[6 * r + c] = (B[3 * r] * J[c] + B[3 * r + 1] * J[6 + c]) + B[3 * r + 2] * J[12 + c];
            const double Bd[3] = {(B[0] * d[0] + B[1] * d[1]) + B[2] * d[2], (B[3] * d[0] + B[4] * d[1]) + B[5] * d[2],
                                  (B[6] * d[0] + B[7] * d[1]) + B[8] * d[2]};
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b)
                    out->JTJ[6 * a + b] += (J[a] * BJ[b] + J[6 + a] * BJ[6 + b]) + J[12 + a] * BJ[12 + b];
                out->JTr[a] += (J[a] * Bd[0] + J[6 + a] * Bd[1]) + J[12 + a] * Bd[2];
            }
            out->r2 += dot3(d, Bd);
        }
    }
}

}  // extern "C"
