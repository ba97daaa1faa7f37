// me_reg.hip — device side of registration_methods 1 / 2 (performICPRegistration, map_eval.cpp:1366-1394):
//   Open3D RegistrationICP(TransformationEstimationPointToPlane) and RegistrationGeneralizedICP [upstream, not in the
//   reference tree; stated Open3D 0.15.1].  What runs here, per cloud or per iteration:
//     * PointCloud::EstimateNormals(KDTreeSearchParamKNN(k)): exact k-NN (one lane per query, nearest-first walk of the
//       sparse octree, the k best kept sorted in LDS) + utility::ComputeCovariance + FastEigen3x3          (k_knn_normals)
//     * InitializePointCloudForGeneralizedICP(epsilon): covariance = Rx diag(eps,1,1) Rx^T                (k_gicp_cov)
//     * PointCloud::Transform on the attributes: n <- R n, C <- R C R^T                                   (k_rotate_attr)
//     * the correspondence + reduction step of one iteration: J^T J, J^T r, sum r^2 over the 1-NN pairs with
//       d2 < max^2 (utility::ComputeJTJandJTr)                                                            (k_lsq_sums)
//   The 6x6 solve and the convergence test stay on the host (icp.py, host/map_eval.cpp).
// 3x3 products accumulate as (a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j; the file is compiled with -ffp-contract=off, so the
// covariance / eigenvector arithmetic follows the CPU path operation by operation (acos / cos may differ by an ulp).
#include <cmath>
#include <cstring>

#include "me_internal.hpp"

namespace me {

namespace {

__device__ __forceinline__ void mat3_mul(const double *a, const double *b, double *o) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void mat3_mul_bt(const double *a, const double *b, double *o) {  // a * b^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o[3 * i + j] = (a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1]) + a[3 * i + 2] * b[3 * j + 2];
}
__device__ __forceinline__ void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// Eigen Matrix3d::inverse(): cofactors, determinant along column 0
__device__ __forceinline__ void inv3(const double *m, double *o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double c10 = m[7] * m[2] - m[8] * m[1], c11 = m[8] * m[0] - m[6] * m[2], c12 = m[6] * m[1] - m[7] * m[0];
    const double c20 = m[1] * m[5] - m[2] * m[4], c21 = m[2] * m[3] - m[0] * m[5], c22 = m[0] * m[4] - m[1] * m[3];
    const double det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
    const double invdet = 1.0 / det;
    o[0] = c00 * invdet;  // out(i,j) = cof(j,i) / det
    o[1] = c10 * invdet;
    o[2] = c20 * invdet;
    o[3] = c01 * invdet;
    o[4] = c11 * invdet;
    o[5] = c21 * invdet;
    o[6] = c02 * invdet;
    o[7] = c12 * invdet;
    o[8] = c22 * invdet;
}

// ---- FastEigen3x3 (Eberly's closed-form symmetric 3x3 eigen-solver as Open3D uses it for normals) ----
__device__ void eigvec0(const double *A, double eval0, double *out) {
    const double row0[3] = {A[0] - eval0, A[1], A[2]};
    const double row1[3] = {A[1], A[4] - eval0, A[5]};
    const double row2[3] = {A[2], A[5], A[8] - eval0};
    double r01[3], r02[3], r12[3];
    cross3(row0, row1, r01);
    cross3(row0, row2, r02);
    cross3(row1, row2, r12);
    const double d0 = dot3(r01, r01), d1 = dot3(r02, r02), d2 = dot3(r12, r12);
    double dmax = d0;
    int imax = 0;
    if (d1 > dmax) {
        dmax = d1;
        imax = 1;
    }
    if (d2 > dmax) imax = 2;
    const double len = sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = (imax == 0 ? r01[k] : (imax == 1 ? r02[k] : r12[k])) / len;
}

__device__ void eigvec1(const double *A, const double *e0, double eval1, double *out) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        const double inv_length = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv_length;
        U[1] = 0;
        U[2] = e0[0] * inv_length;
    } else {
        const double inv_length = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
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
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    double cu = 1, cv = 0;  // out = cu * U - cv * V
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0) {
            if (a00 >= a01) {
                m01 /= m00;
                m00 = 1 / sqrt(1 + m01 * m01);
                m01 *= m00;
            } else {
                m00 /= m01;
                m01 = 1 / sqrt(1 + m00 * m00);
                m00 *= m01;
            }
            cu = m01;
            cv = m00;
        }
    } else {
        if (fmax(a11, a01) > 0) {
            if (a11 >= a01) {
                m01 /= m11;
                m11 = 1 / sqrt(1 + m01 * m01);
                m01 *= m11;
            } else {
                m11 /= m01;
                m01 = 1 / sqrt(1 + m11 * m11);
                m11 *= m01;
            }
            cu = m11;
            cv = m01;
        }
    }
    const bool plain = (cu == 1 && cv == 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = plain ? U[k] : cu * U[k] - cv * V[k];
}

__device__ void fast_eigen3x3(const double *cov, double *out) {
    double A[9];
    double max_coeff = cov[0];
#pragma unroll
    for (int e = 1; e < 9; ++e) max_coeff = fmax(max_coeff, cov[e]);
    if (max_coeff == 0) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) A[e] = cov[e] / max_coeff;
    const double norm = (A[1] * A[1] + A[2] * A[2]) + A[5] * A[5];
    if (norm > 0) {
        const double q = ((A[0] + A[4]) + A[8]) / 3;
        const double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
        const double p = sqrt((((b00 * b00 + b11 * b11) + b22 * b22) + norm * 2) / 6);
        const double c00 = b11 * b22 - A[5] * A[5];
        const double c01 = A[1] * b22 - A[5] * A[2];
        const double c02 = A[1] * A[5] - b11 * A[2];
        const double det = ((b00 * c00 - A[1] * c01) + A[2] * c02) / ((p * p) * p);
        double half_det = det * 0.5;
        half_det = fmin(fmax(half_det, -1.0), 1.0);
        const double angle = acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        const double beta2 = cos(angle) * 2;
        const double beta0 = cos(angle + two_thirds_pi) * 2;
        const double beta1 = -(beta0 + beta2);
        const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
        double ea[3], eb[3];
        if (half_det >= 0) {
            eigvec0(A, ev2, ea);  // ea = evec2
            if (ev2 < ev0 && ev2 < ev1) {
                out[0] = ea[0], out[1] = ea[1], out[2] = ea[2];
                return;
            }
            eigvec1(A, ea, ev1, eb);  // eb = evec1
            if (ev1 < ev0 && ev1 < ev2) {
                out[0] = eb[0], out[1] = eb[1], out[2] = eb[2];
                return;
            }
            cross3(eb, ea, out);  // evec1 x evec2
        } else {
            eigvec0(A, ev0, ea);  // ea = evec0
            if (ev0 < ev1 && ev0 < ev2) {
                out[0] = ea[0], out[1] = ea[1], out[2] = ea[2];
                return;
            }
            eigvec1(A, ea, ev1, eb);
            if (ev1 < ev0 && ev1 < ev2) {
                out[0] = eb[0], out[1] = eb[1], out[2] = eb[2];
                return;
            }
            cross3(ea, eb, out);  // evec0 x evec1
        }
    } else {
        out[0] = out[1] = out[2] = 0;
        if (cov[0] < cov[4] && cov[0] < cov[8]) out[0] = 1;
        else if (cov[4] < cov[0] && cov[4] < cov[8]) out[1] = 1;
        else out[2] = 1;
    }
}

__device__ __forceinline__ double box_lb(const ONode *__restrict__ nd, double qx, double qy, double qz) {
    const double dx = fmax(fmax((double) nd->lo[0] - qx, qx - (double) nd->hi[0]), 0.0);
    const double dy = fmax(fmax((double) nd->lo[1] - qy, qy - (double) nd->hi[1]), 0.0);
    const double dz = fmax(fmax((double) nd->lo[2] - qz, qz - (double) nd->hi[2]), 0.0);
    return (dx * dx + dy * dy) + dz * dz;  // never exceeds the computed d2 of a point inside (boxes rounded outward)
}

constexpr int kKnnBlock = 128;
constexpr int kKnnMax = 40;  // k * 128 lanes * 12 B of LDS <= 60 KB

// ---- k nearest neighbours of every point of a cloud IN that cloud + the normal of their raw-moment covariance ----
// One lane per query (sorted order, so a wave walks neighbouring paths); the k best so far live in LDS, sorted
// ascending by (d2, original index), element j of lane t at [j * blockDim + t] (conflict-free).  The walk is the
// stackless nearest-first one of k_nn1 with the k-th best as the bound (<=: a tie may hold a smaller index).
__global__ void __launch_bounds__(kKnnBlock)
k_knn_normals(const SPoint *__restrict__ sp, long long n, OctView oct, const double *__restrict__ xyz, int k,
              double *__restrict__ normals, int *__restrict__ knn_idx, double *__restrict__ knn_d2) {
    extern __shared__ double s_dyn[];
    double *s_d = s_dyn;                                         // [k][kKnnBlock]
    int *s_i = reinterpret_cast<int *>(s_dyn + k * kKnnBlock);  // [k][kKnnBlock]
    __shared__ long long s_off[kMaxLevels];
    if (threadIdx.x < kMaxLevels) s_off[threadIdx.x] = oct.off[threadIdx.x];
    __syncthreads();
    const int tid = threadIdx.x;
    const long long i = (long long) blockIdx.x * kKnnBlock + tid;
    if (i >= n) return;
    const ONode *__restrict__ nodes = oct.nodes;
    const int L = oct.n_levels - 1;
    const SPoint q = sp[i];
    const double qx = q.x, qy = q.y, qz = q.z;
    int cnt = 0;
    double worst = INFINITY;
    auto consider = [&](double d, int pi) {
        if (cnt == k) {
            if (!(d < worst || (d == worst && pi < s_i[(k - 1) * kKnnBlock + tid]))) return;
        }
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0) {
            const double pd = s_d[(pos - 1) * kKnnBlock + tid];
            const int pidx = s_i[(pos - 1) * kKnnBlock + tid];
            if (d < pd || (d == pd && pi < pidx)) {
                s_d[pos * kKnnBlock + tid] = pd;
                s_i[pos * kKnnBlock + tid] = pidx;
                --pos;
            } else {
                break;
            }
        }
        s_d[pos * kKnnBlock + tid] = d;
        s_i[pos * kKnnBlock + tid] = pi;
        if (cnt < k) ++cnt;
        if (cnt == k) worst = s_d[(k - 1) * kKnnBlock + tid];
    };
    auto scan_leaf = [&](long long leaf) {
        const long long jb = nodes[leaf].begin, je = nodes[leaf + 1].begin;
        for (long long j = jb; j < je; ++j) {
            const SPoint p = sp[j];
            consider(dist2_exact(qx, qy, qz, p.x, p.y, p.z), (int) p.idx);
        }
    };
    if (L == 0) {
        scan_leaf(0);
    } else {
        int l = L;
        long long nd = 0;
        unsigned long long taken_lo = 0, taken_hi = 0;  // one byte of "children already entered" per level 1..8 / 9..16
        for (;;) {
            const ONode *__restrict__ me = nodes + s_off[l] + nd;
            const long long cb = me[0].begin;
            const int cc = (int) (me[1].begin - cb);
            const unsigned int tk = (l <= 8) ? (unsigned int) (taken_lo >> (8 * (l - 1))) & 0xffu
                                             : (unsigned int) (taken_hi >> (8 * (l - 9))) & 0xffu;
            double kd = INFINITY;
            int kc = 8;
            const ONode *__restrict__ ch = nodes + s_off[l - 1] + cb;
            for (int c = 0; c < cc; ++c) {
                if ((tk >> c) & 1u) continue;
                const double lb = box_lb(ch + c, qx, qy, qz);
                if (lb <= worst && lb < kd) {
                    kd = lb;
                    kc = c;
                }
            }
            if (kc >= 8) {
                if (l == L) break;
                nd = me[0].parent;
                ++l;
            } else {
                if (l <= 8) taken_lo |= 1ULL << (8 * (l - 1) + kc);
                else taken_hi |= 1ULL << (8 * (l - 9) + kc);
                if (l == 1) {
                    scan_leaf(s_off[0] + cb + kc);
                } else {
                    --l;
                    nd = cb + kc;
                    if (l <= 8) taken_lo &= ~(0xffULL << (8 * (l - 1)));
                    else taken_hi &= ~(0xffULL << (8 * (l - 9)));
                }
            }
        }
    }
    const long long qi = q.idx;
    if (knn_idx) {
        for (int j = 0; j < k; ++j) {
            knn_idx[qi * k + j] = j < cnt ? s_i[j * kKnnBlock + tid] : -1;
            knn_d2[qi * k + j] = j < cnt ? s_d[j * kKnnBlock + tid] : INFINITY;
        }
    }
    // utility::ComputeCovariance over the neighbours (ascending distance): raw moments / count
    double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (cnt >= 3) {
        double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < cnt; ++j) {
            const long long pj = s_i[j * kKnnBlock + tid];
            const double x = xyz[3 * pj], y = xyz[3 * pj + 1], z = xyz[3 * pj + 2];
            c[0] += x;
            c[1] += y;
            c[2] += z;
            c[3] += x * x;
            c[4] += x * y;
            c[5] += x * z;
            c[6] += y * y;
            c[7] += y * z;
            c[8] += z * z;
        }
        const double m = (double) cnt;
#pragma unroll
        for (int e = 0; e < 9; ++e) c[e] /= m;
        cov[0] = c[3] - c[0] * c[0];
        cov[4] = c[6] - c[1] * c[1];
        cov[8] = c[8] - c[2] * c[2];
        cov[1] = cov[3] = c[4] - c[0] * c[1];
        cov[2] = cov[6] = c[5] - c[0] * c[2];
        cov[5] = cov[7] = c[7] - c[1] * c[2];
    }
    double nv[3];
    fast_eigen3x3(cov, nv);
    if (sqrt(dot3(nv, nv)) == 0.0) {
        nv[0] = nv[1] = 0;
        nv[2] = 1;
    }
    normals[3 * qi] = nv[0];
    normals[3 * qi + 1] = nv[1];
    normals[3 * qi + 2] = nv[2];
}

// InitializePointCloudForGeneralizedICP: C = Rx diag(eps,1,1) Rx^T, Rx = GetRotationFromE1ToX(normal)
__global__ void k_gicp_cov(const double *__restrict__ normals, long long n, double eps, double *__restrict__ cov) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x0 = normals[3 * i], x1 = normals[3 * i + 1], x2 = normals[3 * i + 2];
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!(x0 < -0.99)) {  // (sic) nearly opposite to e1: identity
        const double v[3] = {0.0, -x2, x1};  // e1 x normal
        const double sv[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
        double sv2[9];
        mat3_mul(sv, sv, sv2);
        const double factor = 1 / (1 + x0);
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = (R[e] + sv[e]) + sv2[e] * factor;
    }
    const double Cd[9] = {eps, 0, 0, 0, 1, 0, 0, 0, 1};
    double RC[9], out[9];
    mat3_mul(R, Cd, RC);
    mat3_mul_bt(RC, R, out);
#pragma unroll
    for (int e = 0; e < 9; ++e) cov[9 * i + e] = out[e];
}

struct Rot3 {
    double r[9];
};

__global__ void k_rotate_attr(double *__restrict__ normals, double *__restrict__ cov, long long n, Rot3 R) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (normals) {
        const double v[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) normals[3 * i + r] = (R.r[3 * r] * v[0] + R.r[3 * r + 1] * v[1]) + R.r[3 * r + 2] * v[2];
    }
    if (cov) {
        double C[9], RC[9], out[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) C[e] = cov[9 * i + e];
        mat3_mul(R.r, C, RC);
        mat3_mul_bt(RC, R.r, out);
#pragma unroll
        for (int e = 0; e < 9; ++e) cov[9 * i + e] = out[e];
    }
}

// ---- one linearised step: sums over the correspondences (sorted query order; attributes in original order) ----
constexpr int kLsqD = 29;  // JTJ upper triangle (21) + JTr (6) + r2 + sum_d2

template <int MODE>
__global__ void __launch_bounds__(256)
k_lsq_sums(const SPoint *__restrict__ qsp, const double *__restrict__ d2s, const int *__restrict__ idxs,
           const double *__restrict__ ref_xyz, const double *__restrict__ src_attr, const double *__restrict__ ref_attr,
           long long n, double gate2, double *__restrict__ pd, long long *__restrict__ pc) {
    double s[kLsqD];
#pragma unroll
    for (int k = 0; k < kLsqD; ++k) s[k] = 0;
    long long cnt = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const double d2 = d2s[i];
        if (!(d2 >= 0.0 && d2 < gate2)) continue;  // SearchHybrid(q, max, 1): d2 < max^2 [Open3D, upstream]
        const long long j = idxs[i];
        const SPoint sq = qsp[i];
        const double vs[3] = {sq.x, sq.y, sq.z};
        const double d[3] = {vs[0] - ref_xyz[3 * j], vs[1] - ref_xyz[3 * j + 1], vs[2] - ref_xyz[3 * j + 2]};
        double J[18];  // MODE 1: J[0..5] only
        double B[9], Bd[3];
        if (MODE == 1) {
            const double nt[3] = {ref_attr[3 * j], ref_attr[3 * j + 1], ref_attr[3 * j + 2]};
            cross3(vs, nt, J);
            J[3] = nt[0];
            J[4] = nt[1];
            J[5] = nt[2];
            const double r = dot3(d, nt);
            int t = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) s[t++] += J[a] * J[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) s[21 + a] += J[a] * r;
            s[27] += r * r;
        } else {
            double M[9];
            const long long si = sq.idx;
#pragma unroll
            for (int e = 0; e < 9; ++e) M[e] = ref_attr[9 * j + e] + src_attr[9 * si + e];
            inv3(M, B);
            const double x = vs[0], y = vs[1], z = vs[2];
            const double Jm[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};  // [-skew(vs) | I]
            double BJ[18];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) BJ[6 * r + c] = (B[3 * r] * Jm[c] + B[3 * r + 1] * Jm[6 + c]) + B[3 * r + 2] * Jm[12 + c];
#pragma unroll
            for (int r = 0; r < 3; ++r) Bd[r] = (B[3 * r] * d[0] + B[3 * r + 1] * d[1]) + B[3 * r + 2] * d[2];
            int t = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) s[t++] += (Jm[a] * BJ[b] + Jm[6 + a] * BJ[6 + b]) + Jm[12 + a] * BJ[12 + b];
#pragma unroll
            for (int a = 0; a < 6; ++a) s[21 + a] += (Jm[a] * Bd[0] + Jm[6 + a] * Bd[1]) + Jm[12 + a] * Bd[2];
            s[27] += dot3(d, Bd);
        }
        s[28] += d2;
        ++cnt;
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
#pragma unroll
    for (int k = 0; k < kLsqD; ++k) {
        const double r = block_sum_256(s[k], smd);
        if (threadIdx.x == 0) pd[(long long) blockIdx.x * kLsqD + k] = r;
    }
    const long long rc = block_sum_256_ll(cnt, smi);
    if (threadIdx.x == 0) pc[blockIdx.x] = rc;
}

__global__ void __launch_bounds__(256)
k_lsq_final(const double *__restrict__ pd, const long long *__restrict__ pc, int nblocks, double *__restrict__ out_d,
            long long *__restrict__ out_c) {
    const int k = blockIdx.x;  // 0..kLsqD-1: a double component; kLsqD: the count
    __shared__ double smd[4];
    __shared__ long long smi[4];
    if (k < kLsqD) {
        double s = 0;
        for (int b = threadIdx.x; b < nblocks; b += 256) s += pd[(long long) b * kLsqD + k];
        const double r = block_sum_256(s, smd);
        if (threadIdx.x == 0) out_d[k] = r;
    } else {
        long long s = 0;
        for (int b = threadIdx.x; b < nblocks; b += 256) s += pc[b];
        const long long r = block_sum_256_ll(s, smi);
        if (threadIdx.x == 0) *out_c = r;
    }
}

inline unsigned int grid_for(long long n, int block = 256) { return (unsigned int) ((n + block - 1) / block); }

int need_plain_cloud(me_ctx *ctx, int slot, const char *who, bool need_index) {
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, std::string(who) + ": bad slot");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, std::string(who) + ": cloud not uploaded");
    if (c.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, std::string(who) + ": not available in slab mode");
    if (need_index && !c.index_valid) return ctx->fail(ME_ERR_STATE, std::string(who) + ": cloud has no index");
    if (need_index) ME_TRY(cloud_finish_octree(ctx, slot));
    return ME_OK;
}

}  // namespace

int set_normals(me_ctx *ctx, int slot, const double *normals_host) {
    ME_TRY(need_plain_cloud(ctx, slot, "me_set_normals", false));
    if (!normals_host) return ctx->fail(ME_ERR_ARG, "me_set_normals: normals is NULL");
    Cloud &c = ctx->cloud[slot];
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    ME_CHECK(ctx, c.normals.ensure((size_t) c.n * 24));
    ME_TRY(copy_h2d(ctx, c.normals.p, normals_host, (size_t) c.n * 24));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.have_normals = true;
    c.have_cov = false;
    return ME_OK;
}

int get_normals(me_ctx *ctx, int slot, double *normals_host) {
    ME_TRY(need_plain_cloud(ctx, slot, "me_get_normals", false));
    Cloud &c = ctx->cloud[slot];
    if (!c.have_normals) return ctx->fail(ME_ERR_STATE, "me_get_normals: the cloud has no normals");
    if (!normals_host) return ctx->fail(ME_ERR_ARG, "me_get_normals: normals is NULL");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    ME_TRY(copy_d2h(ctx, normals_host, c.normals.p, (size_t) c.n * 24));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int estimate_normals(me_ctx *ctx, int slot, int knn, double *normals_host, int32_t *knn_idx_host, double *knn_d2_host) {
    ME_TRY(need_plain_cloud(ctx, slot, "me_estimate_normals", true));
    if (knn < 1 || knn > kKnnMax) return ctx->fail(ME_ERR_ARG, "me_estimate_normals: knn must be in [1, 40]");
    if ((knn_idx_host == nullptr) != (knn_d2_host == nullptr))
        return ctx->fail(ME_ERR_ARG, "me_estimate_normals: knn_idx and knn_d2 go together");
    Cloud &c = ctx->cloud[slot];
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = c.n;
    ME_CHECK(ctx, c.normals.ensure((size_t) n * 24));
    int *d_idx = nullptr;
    double *d_d2 = nullptr;
    DevBuf buf_idx, buf_d2;  // released on return: n * knn * 12 bytes would otherwise stay with the context's scratch
    if (knn_idx_host) {
        ME_CHECK(ctx, buf_idx.ensure((size_t) n * knn * 4));
        ME_CHECK(ctx, buf_d2.ensure((size_t) n * knn * 8));
        d_idx = buf_idx.as<int>();
        d_d2 = buf_d2.as<double>();
    }
    {
        TimerScope ts(ctx, "normals");
        const size_t lds = (size_t) knn * kKnnBlock * 12;
        hipLaunchKernelGGL(k_knn_normals, dim3(grid_for(n, kKnnBlock)), dim3(kKnnBlock), lds, ctx->stream, c.sp.as<SPoint>(), n,
                           c.oct, c.xyz.as<double>(), knn, c.normals.as<double>(), d_idx, d_d2);
    }
    ME_CHECK(ctx, hipGetLastError());
    if (normals_host)
        ME_TRY(copy_d2h(ctx, normals_host, c.normals.p, (size_t) n * 24));
    if (knn_idx_host) {
        ME_TRY(copy_d2h(ctx, knn_idx_host, d_idx, (size_t) n * knn * 4));
        ME_TRY(copy_d2h(ctx, knn_d2_host, d_d2, (size_t) n * knn * 8));
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.have_normals = true;
    c.have_cov = false;
    return ME_OK;
}

int gicp_covariances(me_ctx *ctx, int slot, double epsilon, double *cov_host) {
    ME_TRY(need_plain_cloud(ctx, slot, "me_gicp_covariances", true));
    if (!(epsilon > 0)) return ctx->fail(ME_ERR_ARG, "me_gicp_covariances: epsilon must be > 0");
    Cloud &c = ctx->cloud[slot];
    // "Compute covariances the same way is done in the original GICP paper": normals from the 20 nearest neighbours
    if (!c.have_normals) ME_TRY(estimate_normals(ctx, slot, 20, nullptr, nullptr, nullptr));
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = c.n;
    ME_CHECK(ctx, c.cov.ensure((size_t) n * 72));
    {
        TimerScope ts(ctx, "normals");
        hipLaunchKernelGGL(k_gicp_cov, dim3(grid_for(n)), dim3(256), 0, ctx->stream, c.normals.as<double>(), n, epsilon,
                           c.cov.as<double>());
    }
    if (cov_host) ME_TRY(copy_d2h(ctx, cov_host, c.cov.p, (size_t) n * 72));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    c.have_cov = true;
    return ME_OK;
}

int get_covariances(me_ctx *ctx, int slot, double *cov_host) {
    ME_TRY(need_plain_cloud(ctx, slot, "me_get_covariances", false));
    Cloud &c = ctx->cloud[slot];
    if (!c.have_cov) return ctx->fail(ME_ERR_STATE, "me_get_covariances: the cloud has no covariances");
    if (!cov_host) return ctx->fail(ME_ERR_ARG, "me_get_covariances: cov is NULL");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    ME_TRY(copy_d2h(ctx, cov_host, c.cov.p, (size_t) c.n * 72));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

// called by cloud_transform: the attributes follow the points (Open3D PointCloud::Transform)
int rotate_attributes(me_ctx *ctx, int slot, const double *T) {
    Cloud &c = ctx->cloud[slot];
    if (!c.have_normals && !c.have_cov) return ME_OK;
    Rot3 R;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) R.r[3 * r + k] = T[4 * r + k];
    hipLaunchKernelGGL(k_rotate_attr, dim3(grid_for(c.n)), dim3(256), 0, ctx->stream,
                       c.have_normals ? c.normals.as<double>() : nullptr, c.have_cov ? c.cov.as<double>() : nullptr, c.n, R);
    return ME_OK;
}

int icp_lsq_sums(me_ctx *ctx, int qslot, int mode, double max_distance, me_icp_lsq *out) {
    if (qslot < 0 || qslot > 1 || !out || !(max_distance > 0) || (mode != ME_ICP_POINT_TO_PLANE && mode != ME_ICP_GENERALIZED))
        return ctx->fail(ME_ERR_ARG, "me_icp_lsq_sums: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "me_icp_lsq_sums: call me_nn1(query_slot, ref_slot) first");
    if (q.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_icp_lsq_sums: not available in slab mode");
    Cloud &r = ctx->cloud[q.nn_ref_slot];
    if (mode == ME_ICP_POINT_TO_PLANE && !r.have_normals)
        return ctx->fail(ME_ERR_STATE, "me_icp_lsq_sums: point-to-plane needs normals on the target cloud "
                                       "(me_set_normals / me_estimate_normals)");  // Open3D raises the same complaint
    if (mode == ME_ICP_GENERALIZED && (!r.have_cov || !q.have_cov))
        return ctx->fail(ME_ERR_STATE, "me_icp_lsq_sums: generalized ICP needs me_gicp_covariances on both clouds");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    // me_set_shard: this rank's share [b, e) of the sorted queries (the one me_nn1 searched); the sums are additive, the caller
    // all-reduces them (host/map_eval_dist.cpp: the multi-GPU registration loop)
    long long sb, se;
    ctx->shard_range(q.n, sb, se);
    const long long n = se - sb;
    const int nb = (int) std::max<long long>(1, std::min<long long>(1024, (n + 255) / 256));
    const size_t bytes_d = (size_t) (nb + 1) * kLsqD * 8;
    ME_CHECK(ctx, ctx->red.ensure(bytes_d + (size_t) (nb + 1) * 8));
    double *pd = ctx->red.as<double>();
    long long *pc = reinterpret_cast<long long *>(ctx->red.as<char>() + bytes_d);
    {
        TimerScope ts(ctx, "icp");
        if (mode == ME_ICP_POINT_TO_PLANE)
            hipLaunchKernelGGL(k_lsq_sums<1>, dim3(nb), dim3(256), 0, ctx->stream, q.sp.as<SPoint>() + sb, q.nn_d2.as<double>() + sb,
                               q.nn_idx.as<int>() + sb, r.xyz.as<double>(), (const double *) nullptr, r.normals.as<double>(), n,
                               max_distance * max_distance, pd, pc);
        else
            hipLaunchKernelGGL(k_lsq_sums<2>, dim3(nb), dim3(256), 0, ctx->stream, q.sp.as<SPoint>() + sb, q.nn_d2.as<double>() + sb,
                               q.nn_idx.as<int>() + sb, r.xyz.as<double>(), q.cov.as<double>(), r.cov.as<double>(), n,
                               max_distance * max_distance, pd, pc);
        hipLaunchKernelGGL(k_lsq_final, dim3(kLsqD + 1), dim3(256), 0, ctx->stream, pd, pc, nb, pd + (size_t) nb * kLsqD, pc + nb);
    }
    double hd[kLsqD];
    long long hc = 0;
    ME_TRY(copy_d2h(ctx, hd, pd + (size_t) nb * kLsqD, sizeof(hd)));
    ME_TRY(copy_d2h(ctx, &hc, pc + nb, 8));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    out->n_corr = hc;
    out->n_source = q.n;
    int t = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            out->JTJ[6 * a + b] = hd[t];
            out->JTJ[6 * b + a] = hd[t];
            ++t;
        }
    for (int a = 0; a < 6; ++a) out->JTr[a] = hd[21 + a];
    out->r2 = hd[27];
    out->sum_d2 = hd[28];
    return ME_OK;
}

}  // namespace me
