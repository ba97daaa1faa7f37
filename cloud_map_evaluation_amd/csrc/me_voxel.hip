// me_voxel.hip — voxel Gaussians, the Gaussian "Wasserstein" term, AWD, CDF and SCS.
//
//   VoxelCalculator::buildVoxelMap + computeVoxelEntropy   voxel_calculator.cpp:21-56, 97-113, 241-245
//   VoxelCalculator::updateVoxelMap(gt)                    voxel_calculator.cpp:142-172
//   computeWassersteinDistanceGaussian                     voxel_calculator.cpp:115-140
//   MapEval::calculateVMD (AWD mean, CDF, SCS)             map_eval.cpp:240-390
//
// The reference inserts every point into an unordered_map with an XOR hash (99 % of its AWD stage time) and updates
// a streaming Welford mean/M2.  Here: pack floor(p/voxel) into a 63-bit key and reduce the Morton-sorted cloud run by run
// (see "voxel statistics straight from the Morton-sorted cloud" below): two-pass mean / M2.  Tables come out in ascending
// (ix,iy,iz) order; est/gt are joined by binary search, W is one voxel pair per lane, SCS is one wavefront per
// voxel over the (2r+1)^3 stencil with hash probes.
#include <cmath>

#include "me_internal.hpp"
#include "me_vox_rows.hpp"

namespace me {

constexpr unsigned long long kVoxSentinel = 0x7fffffffffffffffULL;  // sorts after every real key

__device__ __forceinline__ double wave_allsum(double v);  // xor butterfly: every lane gets the (fixed-tree) total

// getVoxelIndex (voxel_calculator.cpp:241-245): floor(x / voxel_size) — IEEE division, not a reciprocal multiply
__device__ __forceinline__ unsigned long long voxel_key_of(double x, double y, double z, double vs, const SlabView &slab,
                                                           int *__restrict__ err) {
    if (!slab_owned(slab, x, y, z)) return kVoxSentinel;  // slab mode: halo points belong to another rank
    const double fx = floor(x / vs), fy = floor(y / vs), fz = floor(z / vs);
    const double lim = (double) (kKeyBias - 16);
    if (!(fabs(fx) < lim && fabs(fy) < lim && fabs(fz) < lim)) {
        *err = 1;
        return 0;
    }
    return pack_key((int) fx, (int) fy, (int) fz);
}

// ---- voxel statistics straight from the Morton-sorted cloud ----------------------------------------------------
// A voxel is large next to the search cell, so curve-consecutive points mostly share their voxel: every wavefront
// (64 consecutive sorted points) splits into a few RUNS of equal key.  Pass 1 reduces (n, sum p) per run with a
// segmented butterfly and emits one record per run; the records (about n/60 of them, not n) are radix-sorted by key,
// one wavefront per voxel adds its records up in that fixed order -> mean; pass 2 emits sum (p-mean)(p-mean)^T per run
// the same way.  No per-point sort, no gather through a permutation, and the work per voxel no longer depends on its
// population (one wavefront used to walk a 100 k-point voxel alone).  Deterministic: fixed trees, stable sort.
struct RunLane {
    unsigned long long key;
    bool valid, head;
    int run_local;  // index of the lane's run inside the wave
    int seg;        // id used by the segmented butterfly (unique per run, distinct for invalid lanes)
};

__device__ __forceinline__ RunLane wave_runs(const SPoint *__restrict__ sp, long long i, long long n, double vs,
                                            const SlabView &slab, int *__restrict__ err, int lane, double &x, double &y,
                                            double &z) {
    RunLane r;
    r.valid = i < n;
    x = y = z = 0.0;
    r.key = ~0ULL;
    if (r.valid) {
        const SPoint p = sp[i];
        x = p.x;
        y = p.y;
        z = p.z;
        r.key = voxel_key_of(x, y, z, vs, slab, err);
    }
    const unsigned long long prev = __shfl_up(r.key, 1, 64);
    r.head = r.valid && (lane == 0 || r.key != prev);
    const unsigned long long hm = __ballot(r.head);
    r.run_local = __popcll(hm & ((2ULL << lane) - 1ULL)) - 1;
    r.seg = r.valid ? r.run_local : 64 + lane;
    return r;
}

__global__ void __launch_bounds__(256)
k_vox_count_runs(const SPoint *__restrict__ sp, long long n, double vs, SlabView slab, unsigned int *__restrict__ wave_runs_out,
                 int *__restrict__ err) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double x, y, z;
    const RunLane r = wave_runs(sp, i, n, vs, slab, err, lane, x, y, z);
    const unsigned long long hm = __ballot(r.head);
    if (lane == 0 && r.valid) wave_runs_out[i >> 6] = (unsigned int) __popcll(hm);
}

__global__ void __launch_bounds__(256)
k_vox_pass1(const SPoint *__restrict__ sp, long long n, double vs, SlabView slab, const unsigned int *__restrict__ wave_off,
            unsigned long long *__restrict__ rec_key, unsigned int *__restrict__ rec_iota, int *__restrict__ rec_n,
            double *__restrict__ rec_sum, int *__restrict__ err) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double x, y, z;
    const RunLane r = wave_runs(sp, i, n, vs, slab, err, lane, x, y, z);
    int cnt;
    double sx, sy, sz;
    if (__ballot(r.head) == 1ULL) {  // (wave-uniform) the wavefront is one run: plain sums on the vector unit (invalid lanes hold zeros)
        cnt = __popcll(__ballot(r.valid));
        sx = wave_sum_to_lane0(x);
        sy = wave_sum_to_lane0(y);
        sz = wave_sum_to_lane0(z);
    } else {
        cnt = seg_sum_to_head_i(r.valid ? 1 : 0, r.seg, lane);
        sx = seg_sum_to_head(x, r.seg, lane);
        sy = seg_sum_to_head(y, r.seg, lane);
        sz = seg_sum_to_head(z, r.seg, lane);
    }
    if (r.head) {
        const long long rid = (long long) wave_off[i >> 6] + r.run_local;
        rec_key[rid] = r.key;
        rec_iota[rid] = (unsigned int) rid;
        rec_n[rid] = cnt;
        rec_sum[3 * rid] = sx;
        rec_sum[3 * rid + 1] = sy;
        rec_sum[3 * rid + 2] = sz;
    }
}

// one wavefront per voxel: add the run records up (fixed order) -> population and mean; tag the records with the voxel
__global__ void __launch_bounds__(256)
k_vox_mean(const unsigned int *__restrict__ perm_r, const unsigned int *__restrict__ seg_start, long long n_vox,
           const int *__restrict__ rec_n, const double *__restrict__ rec_sum, unsigned int *__restrict__ rec_vox,
           int *__restrict__ vn, double *__restrict__ vmu) {
    const int lane = threadIdx.x & 63;
    const long long v = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vox) return;
    const long long b = seg_start[v], e = seg_start[v + 1];
    long long cnt = 0;
    double sx = 0, sy = 0, sz = 0;
    for (long long j = b + lane; j < e; j += 64) {
        const unsigned int r = perm_r[j];
        cnt += rec_n[r];
        sx += rec_sum[3 * (long long) r];
        sy += rec_sum[3 * (long long) r + 1];
        sz += rec_sum[3 * (long long) r + 2];
        rec_vox[r] = (unsigned int) v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    const double mx = wave_allsum(sx) / (double) cnt, my = wave_allsum(sy) / (double) cnt, mz = wave_allsum(sz) / (double) cnt;
    if (lane == 0) {
        vn[v] = (int) cnt;
        vmu[3 * v] = mx;
        vmu[3 * v + 1] = my;
        vmu[3 * v + 2] = mz;
    }
}

__global__ void __launch_bounds__(256)
k_vox_pass2(const SPoint *__restrict__ sp, long long n, double vs, SlabView slab, const unsigned int *__restrict__ wave_off,
            const unsigned int *__restrict__ rec_vox, const double *__restrict__ vmu, double *__restrict__ rec_m2,
            int *__restrict__ err) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double x, y, z;
    const RunLane r = wave_runs(sp, i, n, vs, slab, err, lane, x, y, z);
    const long long rid = r.valid ? (long long) wave_off[i >> 6] + r.run_local : 0;
    double dx = 0, dy = 0, dz = 0;
    if (r.valid && r.key != kVoxSentinel) {
        const long long v = rec_vox[rid];
        dx = x - vmu[3 * v];
        dy = y - vmu[3 * v + 1];
        dz = z - vmu[3 * v + 2];
    }
    double cxx, cxy, cxz, cyy, cyz, czz;
    if (__ballot(r.head) == 1ULL) {  // (wave-uniform) one run: see k_vox_pass1
        cxx = wave_sum_to_lane0(dx * dx);
        cxy = wave_sum_to_lane0(dx * dy);
        cxz = wave_sum_to_lane0(dx * dz);
        cyy = wave_sum_to_lane0(dy * dy);
        cyz = wave_sum_to_lane0(dy * dz);
        czz = wave_sum_to_lane0(dz * dz);
    } else {
        cxx = seg_sum_to_head(dx * dx, r.seg, lane);
        cxy = seg_sum_to_head(dx * dy, r.seg, lane);
        cxz = seg_sum_to_head(dx * dz, r.seg, lane);
        cyy = seg_sum_to_head(dy * dy, r.seg, lane);
        cyz = seg_sum_to_head(dy * dz, r.seg, lane);
        czz = seg_sum_to_head(dz * dz, r.seg, lane);
    }
    if (r.head) {
        double *o = rec_m2 + 6 * rid;
        o[0] = cxx;
        o[1] = cxy;
        o[2] = cxz;
        o[3] = cyy;
        o[4] = cyz;
        o[5] = czz;
    }
}

__global__ void k_head_flags(const unsigned long long *__restrict__ keys, long long n, unsigned int *__restrict__ flags) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_seg_scatter(const unsigned long long *__restrict__ keys, const unsigned int *__restrict__ flags,
                              const unsigned int *__restrict__ pos, long long n, unsigned long long *__restrict__ seg_key,
                              unsigned int *__restrict__ seg_start) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) {
        seg_key[pos[i]] = keys[i];
        seg_start[pos[i]] = (unsigned int) i;
    }
}
__global__ void k_set_u32v(unsigned int *p, long long i, unsigned int v) { p[i] = v; }

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double det3_rowmajor(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// one wavefront per voxel: M2 = sum of the run records' M2 (fixed order), then the reference's finalisation
__global__ void __launch_bounds__(256)
k_vox_final(const unsigned int *__restrict__ perm_r, const unsigned int *__restrict__ seg_start, long long n_vox,
            const double *__restrict__ rec_m2, const int *__restrict__ vn, int raw, double *__restrict__ vsig,
            double *__restrict__ vent) {
    const int lane = threadIdx.x & 63;
    const long long v = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vox) return;
    const long long b = seg_start[v], e = seg_start[v + 1];
    double cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
    for (long long j = b + lane; j < e; j += 64) {
        const double *m = rec_m2 + 6 * (long long) perm_r[j];
        cxx += m[0];
        cxy += m[1];
        cxz += m[2];
        cyy += m[3];
        cyz += m[4];
        czz += m[5];
    }
    cxx = wave_allsum(cxx); cxy = wave_allsum(cxy); cxz = wave_allsum(cxz);
    cyy = wave_allsum(cyy); cyz = wave_allsum(cyz); czz = wave_allsum(czz);
    if (lane == 0) {
        const long long cnt = vn[v];
        double S[9] = {cxx, cxy, cxz, cxy, cyy, cyz, cxz, cyz, czz};  // M2 = sum (p-mu)(p-mu)^T (Welford's S, :41)
        double ent = 0.0;
        if (cnt > 10 && !raw) {  // (:47); raw = keep M2 undivided (multi-GPU partials)
            const double nm1 = (double) (cnt - 1);
#pragma unroll
            for (int k = 0; k < 9; ++k) S[k] = S[k] / nm1;  // first division (:48)
#pragma unroll
            for (int k = 0; k < 9; ++k) S[k] = S[k] / nm1;  // second division, computeVoxelEntropy (:102)
            const double det = det3_rowmajor(S);
            if (det > 0) {
                const double PI = 3.141592653589793238463;
                ent = 0.5 * log(pow(2 * PI * exp(1.0), 3.0) * det);  // (:109)
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) vsig[9 * v + k] = S[k];
        vent[v] = ent;
    }
}


// ------------------------------------------------------------------------------------------------------------
// ONE pass over the sorted cloud (round 6).  The three-pass build above reads `sp` three times (run counts, per-run sums, per-run
// centred products about the voxel's mean — which only exists after the second pass).  A voxel's CENTRE is known from its key alone:
// every run emits (n, sum d, sum d d^T) with d = p - centre (|d| <= half a voxel diagonal), the records are sorted, one wavefront
// per voxel adds them up and forms   mu = centre + S1 / n,   M2 = S2 - S1 S1^T / n.   The cancellation costs |mu - centre|^2 against
// the matrix's own scale — a few eps; the tests' tolerance is 1e-9 of that scale.  No counting pass either: every row of 64 sorted
// points owns TWO record slots (its first two runs: a row that crosses a voxel face has two, hardly any has more), further runs are
// appended behind the rows' slots with one atomic per such row (a first version appended every record: 780 k atomics on one
// counter, 8 ms per 50 M points; one slot per row and an atomic for every second run: 60 k atomics, still 0.5 ms).  The used slots
// are compacted (flags + a plain scan), the survivors sorted by key; the order inside a voxel is the sorted cloud's (the merge sort
// is stable and the slots are laid out in row order, the appended ones — runs 3.. of their rows — behind: a fixed order too, because
// a record's KEY carries its row and run number below the voxel's bits).
// Same rows, same runs as the three-pass build: keys and populations identical, means and covariances equal to rounding.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_vox_records(const SPoint *__restrict__ sp, long long n, VoxPack vp, SlabView slab, unsigned long long *__restrict__ rec_key,
              int *__restrict__ rec_n, double *__restrict__ rec_s, unsigned int *__restrict__ rec_count, unsigned int n_rows,
              unsigned int region_size, int *__restrict__ err) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    double x = 0, y = 0, z = 0;
    if (valid) {
        const SPoint p = sp[i];
        x = p.x;
        y = p.y;
        z = p.z;
    }
    vox_emit_row(valid, i, x, y, z, vp, slab, threadIdx.x & 63, rec_key, rec_n, rec_s, rec_count, n_rows, region_size, err);
}

// the fullest overflow region -> cnt[1] (the host compares it with the region size)
__global__ void __launch_bounds__(256) k_vox_region_max(unsigned int *__restrict__ cnt) {
    unsigned int m = 0;
    for (unsigned int i = threadIdx.x; i < kVoxRegions; i += 256) m = max(m, cnt[2 + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int) __shfl_xor((int) m, o, 64));
    __shared__ unsigned int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) cnt[1] = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
}

__global__ void k_used_flags(const unsigned long long *__restrict__ keys, long long n, unsigned int *__restrict__ flags) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = keys[i] != kVoxEmptySlot ? 1u : 0u;
}
// the used slots, in slot order: (key, slot index) pairs for the sort
__global__ void k_compact_used(const unsigned long long *__restrict__ keys, const unsigned int *__restrict__ flags,
                               const unsigned int *__restrict__ pos, long long n, unsigned long long *__restrict__ ckey,
                               unsigned int *__restrict__ cidx) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) {
        ckey[pos[i]] = keys[i];
        cidx[pos[i]] = (unsigned int) i;
    }
}

__global__ void k_head_flags_shifted(const unsigned long long *__restrict__ keys, long long n, int shift, unsigned int *__restrict__ flags) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
}

// one wavefront per voxel: the records of the voxel in sorted order (= the sorted cloud's order) -> n, mu, M2, and the reference's
// finalisation (as k_vox_final)
__global__ void __launch_bounds__(256)
k_vox_reduce(const unsigned long long *__restrict__ skey, const unsigned int *__restrict__ perm_r, const unsigned int *__restrict__ seg_start,
             long long n_vox, VoxPack vp, const int *__restrict__ rec_n, const double *__restrict__ rec_s, int raw,
             unsigned long long *__restrict__ vkey, int *__restrict__ vn, double *__restrict__ vmu, double *__restrict__ vsig,
             double *__restrict__ vent) {
    const int lane = threadIdx.x & 63;
    const long long v = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vox) return;
    const long long b = seg_start[v], e = seg_start[v + 1];
    long long cnt = 0;
    double a[kVoxRec];
#pragma unroll
    for (int k = 0; k < kVoxRec; ++k) a[k] = 0.0;
    for (long long j = b + lane; j < e; j += 64) {
        const long long r = perm_r[j];
        cnt += rec_n[r];
#pragma unroll
        for (int k = 0; k < kVoxRec; ++k) a[k] += rec_s[kVoxRec * r + k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
    for (int k = 0; k < kVoxRec; ++k) a[k] = wave_allsum(a[k]);
    if (lane != 0) return;
    const unsigned long long ck = skey[b] >> vp.pos_bits;
    const int ix = (int) (ck >> (vp.bits_y + vp.bits_z)) + vp.min_x;
    const int iy = (int) ((ck >> vp.bits_z) & ((1ULL << vp.bits_y) - 1ULL)) + vp.min_y;
    const int iz = (int) (ck & ((1ULL << vp.bits_z) - 1ULL)) + vp.min_z;
    vkey[v] = pack_key(ix, iy, iz);
    vn[v] = (int) cnt;
    const double nn = (double) cnt;
    const double mx = a[0] / nn, my = a[1] / nn, mz = a[2] / nn;  // mean offset from the voxel's centre
    vmu[3 * v] = ((double) ix + 0.5) * vp.vs + mx;
    vmu[3 * v + 1] = ((double) iy + 0.5) * vp.vs + my;
    vmu[3 * v + 2] = ((double) iz + 0.5) * vp.vs + mz;
    // M2 = sum (p - mu)(p - mu)^T = S2 - S1 S1^T / n (Welford's S, voxel_calculator.cpp:41)
    const double cxx = a[3] - a[0] * mx, cxy = a[4] - a[0] * my, cxz = a[5] - a[0] * mz;
    const double cyy = a[6] - a[1] * my, cyz = a[7] - a[1] * mz, czz = a[8] - a[2] * mz;
    double S[9] = {cxx, cxy, cxz, cxy, cyy, cyz, cxz, cyz, czz};
    double ent = 0.0;
    if (cnt > 10 && !raw) {  // (:47); raw = keep M2 undivided (multi-GPU partials)
        const double nm1 = (double) (cnt - 1);
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] = S[k] / nm1;  // first division (:48)
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] = S[k] / nm1;  // second division, computeVoxelEntropy (:102)
        const double det = det3_rowmajor(S);
        if (det > 0) {
            const double PI = 3.141592653589793238463;
            ent = 0.5 * log(pow(2 * PI * exp(1.0), 3.0) * det);  // (:109)
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) vsig[9 * v + k] = S[k];
    vent[v] = ent;
}

// ---- join est x gt on the packed key (both ascending) ----
__global__ void k_join(const unsigned long long *__restrict__ ekey, const int *__restrict__ en, long long Ve,
                       const unsigned long long *__restrict__ gkey, const int *__restrict__ gn, long long Vg, int min_pts,
                       int *__restrict__ gi_out, unsigned int *__restrict__ match, unsigned int *__restrict__ active) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ve) return;
    const unsigned long long k = ekey[i];
    long long lo = 0, hi = Vg;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (gkey[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < Vg && gkey[lo] == k;
    gi_out[i] = found ? (int) lo : -1;
    active[i] = found ? 1u : 0u;  // active == 1 (voxel_calculator.cpp:153)
    match[i] = (found && en[i] >= min_pts && gn[lo] >= min_pts) ? 1u : 0u;  // (map_eval.cpp:274-281)
}

// ---- 3x3 helpers for computeWassersteinDistanceGaussian ----
__device__ __forceinline__ void jacobi_rot(double *a, double *V, int p, int q) {
    const double apq = a[3 * p + q];
    if (apq == 0.0) return;
    const double app = a[3 * p + p], aqq = a[3 * q + q];
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double akp = a[3 * k + p], akq = a[3 * k + q];
        a[3 * k + p] = c * akp - s * akq;
        a[3 * k + q] = s * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double apk = a[3 * p + k], aqk = a[3 * q + k];
        a[3 * p + k] = c * apk - s * aqk;
        a[3 * q + k] = s * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = c * vkp - s * vkq;
        V[3 * k + q] = s * vkp + c * vkq;
    }
}

// sigma_stored/(n-1), symmetrise, eigen-clamp at 1e-6, rebuild (voxel_calculator.cpp:119-125)
__device__ void regularize_sigma(const double *__restrict__ stored, int n, double *out) {
    if (n > 1) {
        double s[9], a[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] = stored[k] / (double) (n - 1);  // third division (:120)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[3 * r + c] = (s[3 * r + c] + s[3 * c + r]) / 2;  // (:121)
        for (int sweep = 0; sweep < 64; ++sweep) {
            const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
            const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
            if (off <= 1e-32 * diag || off == 0.0) break;
            jacobi_rot(a, V, 0, 1);
            jacobi_rot(a, V, 0, 2);
            jacobi_rot(a, V, 1, 2);
        }
        const double ev[3] = {fmax(a[0], 1e-6), fmax(a[4], 1e-6), fmax(a[8], 1e-6)};  // cwiseMax(1e-6) (:123)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc += V[3 * r + k] * ev[k] * V[3 * c + k];
                out[3 * r + c] = acc;
            }
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = (k % 4 == 0) ? 1.0 : 0.0;
    }
}

__device__ __forceinline__ void chol3(const double *a, double *L) {  // Eigen LLT, lower, info() unchecked (:136-137)
#pragma unroll
    for (int k = 0; k < 9; ++k) L[k] = 0.0;
    L[0] = sqrt(a[0]);
    L[3] = a[3] / L[0];
    L[6] = a[6] / L[0];
    L[4] = sqrt(a[4] - L[3] * L[3]);
    L[7] = (a[7] - L[6] * L[3]) / L[4];
    L[8] = sqrt(a[8] - L[6] * L[6] - L[7] * L[7]);
}

__device__ double w2_gaussian(const double *mu1, const double *sig1, int n1, const double *mu2, const double *sig2, int n2) {
    double s1[9], s2[9];
    regularize_sigma(sig1, n1, s1);
    regularize_sigma(sig2, n2, s2);
    double md = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) md += (mu1[d] - mu2[d]) * (mu1[d] - mu2[d]);
    const double tr_sum = (s1[0] + s2[0]) + (s1[4] + s2[4]) + (s1[8] + s2[8]);
    double L1[9], T[9], M[9], L[9];
    chol3(s1, L1);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += L1[3 * r + k] * s2[3 * k + c];
            T[3 * r + c] = acc;
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += T[3 * r + k] * L1[3 * c + k];
            M[3 * r + c] = acc;  // L1 * sigma2 * L1^T (:136)
        }
    chol3(M, L);                                                    // (:137)
    const double distance = md + tr_sum - 2 * (L[0] + L[4] + L[8]);  // (:138)
    return sqrt(fmax(0.0, distance));                                // (:139)
}

__global__ void k_w2(const unsigned int *__restrict__ match, const unsigned int *__restrict__ mpos, const int *__restrict__ gi,
                     long long Ve, const unsigned long long *__restrict__ ekey, const int *__restrict__ en,
                     const double *__restrict__ emu, const double *__restrict__ esig, const int *__restrict__ gn,
                     const double *__restrict__ gmu, const double *__restrict__ gsig, double vs,
                     unsigned long long *__restrict__ mkey, double *__restrict__ mw, double *__restrict__ rows) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ve || !match[i]) return;
    const long long m = mpos[i];
    const long long g = gi[i];
    // computeWassersteinDistanceGaussian(gt_voxel, est_voxel) — gt first (map_eval.cpp:284)
    const double w = w2_gaussian(gmu + 3 * g, gsig + 9 * g, gn[g], emu + 3 * i, esig + 9 * i, en[i]);
    mkey[m] = ekey[i];
    mw[m] = w;
    if (rows) {  // 27 columns of voxel_errors.txt (map_eval.cpp:292-302)
        double *r = rows + 27 * m;
        int kx, ky, kz;
        unpack_key(ekey[i], kx, ky, kz);
        r[0] = (double) kx * vs; r[1] = (double) ky * vs; r[2] = (double) kz * vs;
        r[3] = ((double) kx + 1.0) * vs; r[4] = ((double) ky + 1.0) * vs; r[5] = ((double) kz + 1.0) * vs;
        r[6] = emu[3 * i]; r[7] = emu[3 * i + 1]; r[8] = emu[3 * i + 2];
        r[9] = w;
        r[10] = (double) gn[g];
        r[11] = (double) en[i];
        const double *es = esig + 9 * i, *gs = gsig + 9 * g;
        r[12] = es[0]; r[13] = es[1]; r[14] = es[2]; r[15] = es[4]; r[16] = es[5]; r[17] = es[8];
        r[18] = gmu[3 * g]; r[19] = gmu[3 * g + 1]; r[20] = gmu[3 * g + 2];
        r[21] = gs[0]; r[22] = gs[1]; r[23] = gs[2]; r[24] = gs[4]; r[25] = gs[5]; r[26] = gs[8];
    }
}

__global__ void k_hash_insert_keys(const unsigned long long *__restrict__ keys, long long n,
                                   unsigned long long *__restrict__ hkeys, unsigned int *__restrict__ hvals, unsigned int mask) {
    const long long c = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const unsigned long long key = keys[c];
    unsigned int s = (unsigned int) hash_u64(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&hkeys[s], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) {
            hvals[s] = (unsigned int) c;
            return;
        }
        s = (s + 1) & mask;
    }
}

// SCS: one wavefront (= one 64-thread block) per W-voxel; dynamic LDS holds the stencil's W values (< 0 = absent)
__global__ void __launch_bounds__(64)
k_scs(const unsigned long long *__restrict__ mkey, const double *__restrict__ mw, long long M,
      const unsigned long long *__restrict__ hkeys, const unsigned int *__restrict__ hvals, unsigned int hmask, int radius,
      double *__restrict__ scs_i, unsigned int *__restrict__ has_nb) {
    extern __shared__ __align__(16) double wbuf[];
    const long long v = blockIdx.x;
    if (v >= M) return;
    const int lane = threadIdx.x;
    const int side = 2 * radius + 1, total = side * side * side, center = (total - 1) / 2;
    int kx, ky, kz;
    unpack_key(mkey[v], kx, ky, kz);
    double s = 0;
    int c = 0;
    for (int o = lane; o < total; o += 64) {
        double w = -1.0;
        if (o != center) {  // getNeighborIndices skips the centre voxel (voxel_calculator.cpp:12-13)
            const int dx = o / (side * side) - radius, dy = (o / side) % side - radius, dz = o % side - radius;
            const int nx = kx + dx, ny = ky + dy, nz = kz + dz;
            if (abs(nx) < kKeyBias && abs(ny) < kKeyBias && abs(nz) < kKeyBias) {
                const int hi = hash_lookup(hkeys, hvals, hmask, pack_key(nx, ny, nz));
                if (hi >= 0) {
                    w = mw[hi];
                    s += w;
                    ++c;
                }
            }
        }
        wbuf[o] = w;
    }
    const double tot = wave_allsum(s);
    int cnt = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __syncthreads();
    if (cnt == 0) {  // neighbor_ws_distances.empty() (map_eval.cpp:372)
        if (lane == 0) {
            scs_i[v] = 0.0;
            has_nb[v] = 0u;
        }
        return;
    }
    const double mean = tot / (double) cnt;  // (:373-374)
    double var = 0;
    for (int o = lane; o < total; o += 64) {
        const double w = wbuf[o];
        if (w >= 0.0) var += (w - mean) * (w - mean);  // (:376-378)
    }
    var = wave_allsum(var) / (double) cnt;  // population variance (:379)
    if (lane == 0) {
        scs_i[v] = sqrt(var) / mean;  // (:380-381); inf/NaN propagate when mean == 0, as the reference
        has_nb[v] = 1u;
    }
}

// deterministic sum of n doubles (optionally masked) by one 256-thread block
__global__ void __launch_bounds__(256)
k_sum_masked(const double *__restrict__ x, const unsigned int *__restrict__ mask, long long n, double *__restrict__ out_s,
             long long *__restrict__ out_c) {
    double s = 0;
    long long c = 0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        if (!mask || mask[i]) {
            s += x[i];
            ++c;
        }
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double rs = block_sum_256(s, smd);
    const long long rc = block_sum_256_ll(c, smi);
    if (threadIdx.x == 0) {
        *out_s = rs;
        *out_c = rc;
    }
}
__global__ void __launch_bounds__(256) k_count_u32(const unsigned int *__restrict__ x, long long n, long long *__restrict__ out) {
    long long c = 0;
    for (long long i = threadIdx.x; i < n; i += 256) c += x[i];
    __shared__ long long smi[4];
    const long long rc = block_sum_256_ll(c, smi);
    if (threadIdx.x == 0) *out = rc;
}

__global__ void k_unpack_keys(const unsigned long long *__restrict__ k, long long n, int *__restrict__ out) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    unpack_key(k[i], x, y, z);
    out[3 * i] = x;
    out[3 * i + 1] = y;
    out[3 * i + 2] = z;
}

// ---- VoxelDownSample (open3d::geometry::PointCloud::VoxelDownSample, map_eval.cpp:38-39) ----
// voxel index = floor((p - (min_bound - vs/2)) / vs)  [Open3D, upstream]; keys are non-negative here.
__global__ void k_vds_keys(const double *__restrict__ xyz, long long n, double vs, double mx, double my, double mz,
                           unsigned long long *__restrict__ keys, unsigned int *__restrict__ iota, int *__restrict__ err) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double fx = floor((xyz[3 * i] - mx) / vs), fy = floor((xyz[3 * i + 1] - my) / vs), fz = floor((xyz[3 * i + 2] - mz) / vs);
    const double lim = 2097151.0;
    if (!(fx >= 0 && fy >= 0 && fz >= 0 && fx <= lim && fy <= lim && fz <= lim)) {
        *err = 1;
        keys[i] = 0;
    } else {
        keys[i] = ((unsigned long long) fx << 42) | ((unsigned long long) fy << 21) | (unsigned long long) fz;
    }
    iota[i] = (unsigned int) i;
}

// one thread per voxel: sum in ORIGINAL cloud order (the radix sort is stable), divide by the count — exactly the
// AccumulatedPoint arithmetic of Open3D, so every output point is bit-identical to the CPU path
__global__ void k_vds_mean(const double *__restrict__ xyz, const unsigned int *__restrict__ perm,
                           const unsigned int *__restrict__ seg_start, long long n_vox, double *__restrict__ out) {
    const long long v = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vox) return;
    const long long b = seg_start[v], e = seg_start[v + 1];
    double sx = 0, sy = 0, sz = 0;
    for (long long j = b; j < e; ++j) {
        const long long s = perm[j];
        sx += xyz[3 * s];
        sy += xyz[3 * s + 1];
        sz += xyz[3 * s + 2];
    }
    const double cnt = (double) (e - b);
    out[3 * v] = sx / cnt;
    out[3 * v + 1] = sy / cnt;
    out[3 * v + 2] = sz / cnt;
}

static inline unsigned int grid_for(long long n, int block = 256) { return (unsigned int) std::max<long long>(1, (n + block - 1) / block); }

// SCS over a device-resident sparse W table: writes sum(scs_i) and #voxels-with-neighbours to d_sum / d_count
static int scs_device(me_ctx *ctx, const unsigned long long *mkey, const double *mw, long long M, int scs_radius,
                      double *d_sum, long long *d_count) {
    DevBuf hk, hv, scs_d, has_d;
    unsigned long long hsize = 64;
    while (hsize < 2ULL * (unsigned long long) M) hsize <<= 1;
    ME_CHECK(ctx, hk.ensure((size_t) hsize * 8));
    ME_CHECK(ctx, hv.ensure((size_t) hsize * 4));
    ME_CHECK(ctx, scs_d.ensure((size_t) M * 8));
    ME_CHECK(ctx, has_d.ensure((size_t) M * 4));
    ME_CHECK(ctx, hipMemsetAsync(hk.p, 0xFF, (size_t) hsize * 8, ctx->stream));
    hipLaunchKernelGGL(k_hash_insert_keys, dim3(grid_for(M)), dim3(256), 0, ctx->stream, mkey, M, hk.as<unsigned long long>(),
                       hv.as<unsigned int>(), (unsigned int) (hsize - 1));
    const int side = 2 * scs_radius + 1;
    const size_t lds = (size_t) side * side * side * 8;
    {
        TimerScope ts(ctx, "scs");
        hipLaunchKernelGGL(k_scs, dim3((unsigned int) M), dim3(64), lds, ctx->stream, mkey, mw, M, hk.as<unsigned long long>(),
                           hv.as<unsigned int>(), (unsigned int) (hsize - 1), scs_radius, scs_d.as<double>(),
                           has_d.as<unsigned int>());
    }
    hipLaunchKernelGGL(k_sum_masked, dim3(1), dim3(256), 0, ctx->stream, scs_d.as<double>(), has_d.as<unsigned int>(), M, d_sum,
                       d_count);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the local buffers die with this frame
    return ME_OK;
}

__global__ void k_w2_batch(const double *__restrict__ mu1, const double *__restrict__ s1, const int *__restrict__ n1,
                           const double *__restrict__ mu2, const double *__restrict__ s2, const int *__restrict__ n2,
                           long long count, double *__restrict__ w) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    w[i] = w2_gaussian(mu1 + 3 * i, s1 + 9 * i, n1[i], mu2 + 3 * i, s2 + 9 * i, n2[i]);
}

__global__ void k_pack_keys(const int *__restrict__ k3, long long n, unsigned long long *__restrict__ out, int *__restrict__ err) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = k3[3 * i], y = k3[3 * i + 1], z = k3[3 * i + 2];
    if (abs(x) >= kKeyBias - 16 || abs(y) >= kKeyBias - 16 || abs(z) >= kKeyBias - 16) *err = 1;
    out[i] = pack_key(x, y, z);
}

int w2_batch(me_ctx *ctx, const double *mu1, const double *sigma1, const int32_t *n1, const double *mu2, const double *sigma2,
             const int32_t *n2, long long count, double *w) {
    if (count < 0 || (count > 0 && (!mu1 || !sigma1 || !n1 || !mu2 || !sigma2 || !n2 || !w)))
        return ctx->fail(ME_ERR_ARG, "me_w2_batch: bad argument");
    if (count == 0) return ME_OK;
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf b[7];
    const size_t sz[7] = {(size_t) count * 24, (size_t) count * 72, (size_t) count * 4, (size_t) count * 24,
                          (size_t) count * 72, (size_t) count * 4, (size_t) count * 8};
    const void *src[6] = {mu1, sigma1, n1, mu2, sigma2, n2};
    for (int k = 0; k < 7; ++k) ME_CHECK(ctx, b[k].ensure(sz[k]));
    for (int k = 0; k < 6; ++k) ME_TRY(copy_h2d(ctx, b[k].p, src[k], sz[k]));
    {
        TimerScope ts(ctx, "w2");
        hipLaunchKernelGGL(k_w2_batch, dim3(grid_for(count)), dim3(256), 0, ctx->stream, b[0].as<double>(), b[1].as<double>(),
                           b[2].as<int>(), b[3].as<double>(), b[4].as<double>(), b[5].as<int>(), count, b[6].as<double>());
    }
    ME_TRY(copy_d2h(ctx, w, b[6].p, sz[6]));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

int scs_table(me_ctx *ctx, const int32_t *keys, const double *w, long long n, int scs_radius, double *scs) {
    if (n < 0 || !scs || (n > 0 && (!keys || !w))) return ctx->fail(ME_ERR_ARG, "me_scs_table: bad argument");
    if (scs_radius < 1 || scs_radius > 10) return ctx->fail(ME_ERR_ARG, "scs_radius must be in [1, 10]");
    ME_TRACE_POINT(ctx, "awd_scs: enter");
    if (n == 0) {
        *scs = std::nan("");
        return ME_OK;
    }
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf k3, kk, ww;
    ME_CHECK(ctx, k3.ensure((size_t) n * 12));
    ME_CHECK(ctx, kk.ensure((size_t) n * 8));
    ME_CHECK(ctx, ww.ensure((size_t) n * 8));
    ME_CHECK(ctx, ctx->red.ensure(256));
    int *d_err = ctx->red.as<int>();
    double *d_sum = reinterpret_cast<double *>(ctx->red.as<char>() + 64);
    long long *d_cnt = reinterpret_cast<long long *>(ctx->red.as<char>() + 128);
    ME_CHECK(ctx, hipMemsetAsync(d_err, 0, 4, ctx->stream));
    ME_TRY(copy_h2d(ctx, k3.p, keys, (size_t) n * 12));
    ME_TRY(copy_h2d(ctx, ww.p, w, (size_t) n * 8));
    hipLaunchKernelGGL(k_pack_keys, dim3(grid_for(n)), dim3(256), 0, ctx->stream, k3.as<int>(), n, kk.as<unsigned long long>(), d_err);
    ME_TRY(scs_device(ctx, kk.as<unsigned long long>(), ww.as<double>(), n, scs_radius, d_sum, d_cnt));
    int h_err = 0;
    double h_s = 0;
    long long h_c = 0;
    {
        MailGuard mg(ctx);  // (one synchronisation for the batch; destinations are locals of this frame)
        ME_TRY(mail_post(ctx, &h_err, d_err, 4));
        ME_TRY(mail_post(ctx, &h_s, d_sum, 8));
        ME_TRY(mail_post(ctx, &h_c, d_cnt, 8));
        ME_TRY(mg.sync());
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err) return ctx->fail(ME_ERR_ARG, "me_scs_table: voxel index out of range");
    *scs = h_s / (double) h_c;
    return ME_OK;
}


// ------------------------------------------------------------------------------------------------------------
// host side of the one-pass build
static int voxel_build_onepass(me_ctx *ctx, Cloud &c, double voxel_size, bool raw, const VoxPack &vp, int *d_err) {
    const long long n = c.n;
    const SPoint *sp = c.sp.as<SPoint>();
    (void) d_err;
    const long long nw = (n + 63) / 64;  // rows of 64 sorted points: two record slots each, further runs behind them
    DevBuf &kbuf = ctx->tmp[1], &sbuf = ctx->tmp[2], &ibuf = ctx->tmp[3], &mbuf = ctx->tmp[4];  // (tmp[5]: sort / scan scratch)
    const long long cap = vox_record_capacity(n), rsize = vox_region_size(n);
    const long long S = cap;  // every slot is looked at: the rows' two each, then the regions (with holes: compacted below)
    // the records the index build's gather left on the cloud (same rows, same arithmetic: me_vox_rows.hpp), when they are for this
    // voxel size; otherwise the pass below.  A region that overflowed: the three-pass build.
    const bool fused = c.vox_rec_valid && c.vox_rec_pack.vs == voxel_size && c.vox_rec_cap == cap;
    const unsigned long long *slot_key_src = nullptr;
    const int *rec_n_src = nullptr;
    const double *rec_s_src = nullptr;
    unsigned int *cnt = nullptr;
    if (fused) {
        cnt = c.vox_rec_cnt.as<unsigned int>();
        slot_key_src = c.vox_rec_key.as<unsigned long long>();
        rec_n_src = c.vox_rec_n.as<int>();
        rec_s_src = c.vox_rec_s.as<double>();
    } else {
        ME_CHECK(ctx, kbuf.ensure((size_t) cap * 24));      // slot keys | compacted keys | sorted keys
        ME_CHECK(ctx, sbuf.ensure((size_t) cap * 8 * kVoxRec));
        ME_CHECK(ctx, c.vox_rec_cnt.ensure(kVoxCounterBytes));
        cnt = c.vox_rec_cnt.as<unsigned int>();
        ME_CHECK(ctx, hipMemsetAsync(kbuf.p, 0xFF, (size_t) cap * 8, ctx->stream));  // kVoxEmptySlot
        ME_CHECK(ctx, hipMemsetAsync(cnt, 0, kVoxCounterBytes, ctx->stream));
        ME_CHECK(ctx, ibuf.ensure((size_t) cap * 12));      // rec_n | compacted slot indices | sorted slot indices
        TimerScope ts(ctx, "voxel");
        hipLaunchKernelGGL(k_vox_records, dim3(grid_for(n)), dim3(256), 0, ctx->stream, sp, n, vp, c.slab, kbuf.as<unsigned long long>(),
                           ibuf.as<int>(), sbuf.as<double>(), cnt + 2, (unsigned int) nw, (unsigned int) rsize, reinterpret_cast<int *>(cnt));
        slot_key_src = kbuf.as<unsigned long long>();
        rec_n_src = ibuf.as<int>();
        rec_s_src = sbuf.as<double>();
    }
    ME_CHECK(ctx, kbuf.ensure((size_t) cap * 24));
    ME_CHECK(ctx, ibuf.ensure((size_t) cap * 12));
    ME_CHECK(ctx, mbuf.ensure((size_t) cap * 8 + 64));  // flags | positions
    hipLaunchKernelGGL(k_vox_region_max, dim3(1), dim3(256), 0, ctx->stream, cnt);
    {
        unsigned int h2[2] = {0, 0};
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, h2, cnt, 8));
        ME_TRY(mg.sync());
        if (h2[0]) return ctx->fail(ME_ERR_ARG, "voxel index out of range (|floor(p/voxel_size)| must be < 2^20)");
        if ((long long) h2[1] > rsize) return 1;  // (a region overflowed: the caller takes the three-pass build)
    }
    unsigned long long *ckey = kbuf.as<unsigned long long>() + cap, *skey = ckey + cap;
    const unsigned long long *slot_key = slot_key_src;
    const int *rec_n = rec_n_src;
    unsigned int *cidx = ibuf.as<unsigned int>() + cap, *perm_r = cidx + cap;
    unsigned int *flags = mbuf.as<unsigned int>(), *pos = flags + cap;
    // the used slots, compacted in slot order
    hipLaunchKernelGGL(k_used_flags, dim3(grid_for(S)), dim3(256), 0, ctx->stream, slot_key, S, flags);
    ME_TRY(exclusive_scan_u32_plain(ctx, flags, pos, S));
    unsigned int last_pos = 0, last_flag = 0;
    {
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, &last_pos, pos + (S - 1), 4));
        ME_TRY(mail_post(ctx, &last_flag, flags + (S - 1), 4));
        ME_TRY(mg.sync());
    }
    const long long R = (long long) last_pos + last_flag;  // run records
    hipLaunchKernelGGL(k_compact_used, dim3(grid_for(S)), dim3(256), 0, ctx->stream, slot_key, (const unsigned int *) flags,
                       (const unsigned int *) pos, S, ckey, cidx);
    ME_TRY(sort_pairs_merge_u64_u32(ctx, ckey, skey, cidx, perm_r, R));
    hipLaunchKernelGGL(k_head_flags_shifted, dim3(grid_for(R)), dim3(256), 0, ctx->stream, (const unsigned long long *) skey, R, vp.pos_bits, flags);
    ME_TRY(exclusive_scan_u32_plain(ctx, flags, pos, R));
    unsigned long long last_key = 0;
    {
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, &last_pos, pos + (R - 1), 4));
        ME_TRY(mail_post(ctx, &last_flag, flags + (R - 1), 4));
        ME_TRY(mail_post(ctx, &last_key, skey + (R - 1), 8));
        ME_TRY(mg.sync());
    }
    long long V = (long long) last_pos + last_flag;
    const bool has_sentinel = c.slab.axis >= 0 && (last_key >> vp.pos_bits) == vp.sentinel;
    if (has_sentinel) V -= 1;  // the halo points' segment (the last one) is dropped
    ME_CHECK(ctx, c.vox_tmp.ensure((size_t) (V + 2) * 4));
    unsigned int *seg_start = c.vox_tmp.as<unsigned int>();
    ME_CHECK(ctx, c.vox_key.ensure((size_t) (V + 1) * 8));
    ME_CHECK(ctx, c.vox_n.ensure((size_t) std::max<long long>(V, 1) * 4));
    ME_CHECK(ctx, c.vox_mu.ensure((size_t) std::max<long long>(V, 1) * 24));
    ME_CHECK(ctx, c.vox_sigma.ensure((size_t) std::max<long long>(V, 1) * 72));
    ME_CHECK(ctx, c.vox_entropy.ensure((size_t) std::max<long long>(V, 1) * 8));
    // (k_seg_scatter also writes the heads' keys: into ckey, dead since the sort)
    hipLaunchKernelGGL(k_seg_scatter, dim3(grid_for(R)), dim3(256), 0, ctx->stream, (const unsigned long long *) skey, (const unsigned int *) flags,
                       (const unsigned int *) pos, R, ckey, seg_start);
    if (!has_sentinel) hipLaunchKernelGGL(k_set_u32v, dim3(1), dim3(1), 0, ctx->stream, seg_start, V, (unsigned int) R);
    if (V > 0) {
        TimerScope ts(ctx, "voxel");
        hipLaunchKernelGGL(k_vox_reduce, dim3((unsigned int) ((V + 3) / 4)), dim3(256), 0, ctx->stream, (const unsigned long long *) skey,
                           (const unsigned int *) perm_r, (const unsigned int *) seg_start, V, vp, rec_n, rec_s_src,
                           raw ? 1 : 0, c.vox_key.as<unsigned long long>(), c.vox_n.as<int>(), c.vox_mu.as<double>(), c.vox_sigma.as<double>(),
                           c.vox_entropy.as<double>());
    }
    ME_CHECK(ctx, hipGetLastError());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.n_vox = V;
    c.vox_size = voxel_size;
    c.vox_raw = raw;
    c.vox_valid = true;
    return ME_OK;
}

// ------------------------------------------------------------------------------------------------------------
int voxel_build(me_ctx *ctx, int slot, double voxel_size, bool raw) {
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    if (!(voxel_size > 0)) return ctx->fail(ME_ERR_ARG, "voxel_size must be > 0");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "voxel pass: cloud not uploaded");
    if (c.vox_valid && !c.vox_merged && c.vox_size == voxel_size && c.vox_raw == raw) return ME_OK;  // cached
    c.vox_valid = false;
    c.vox_merged = false;
    if (c.n == 0) {  // empty slab
        c.n_vox = 0;
        c.vox_size = voxel_size;
        c.vox_raw = raw;
        c.vox_valid = true;
        return ME_OK;
    }
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    if (!c.index_valid) ME_TRY(cloud_build_index(ctx, slot, c.cell_size_req));
    ME_TRACE_POINT(ctx, "voxel_build: enter");
    const long long n = c.n;
    const long long nw = (n + 63) / 64;
    const SPoint *sp = c.sp.as<SPoint>();
    ME_CHECK(ctx, ctx->red.ensure(64));
    int *d_err = ctx->red.as<int>();
    ME_CHECK(ctx, hipMemsetAsync(d_err, 0, 8, ctx->stream));  // [0] index-range error flag, [1] record counter of the one-pass build
#if ME_TUNE_VOX_ONEPASS
    {
        VoxPack vp;
        if (vox_make_pack(c, voxel_size, n, vp)) {
            const int rc1 = voxel_build_onepass(ctx, c, voxel_size, raw, vp, d_err);
            if (rc1 <= 0) return rc1;  // (1: an overflow region was too small for this cloud — the three-pass build below)
            ME_CHECK(ctx, hipMemsetAsync(d_err, 0, 8, ctx->stream));
        }
    }
#endif
    // --- runs per wavefront -> record offsets ---
    DevBuf &wbuf = ctx->tmp[0];
    ME_CHECK(ctx, wbuf.ensure((size_t) nw * 8));
    unsigned int *wave_runs = wbuf.as<unsigned int>(), *wave_off = wave_runs + nw;
    {
        TimerScope ts(ctx, "voxel");
        hipLaunchKernelGGL(k_vox_count_runs, dim3(grid_for(n)), dim3(256), 0, ctx->stream, sp, n, voxel_size, c.slab, wave_runs, d_err);
    }
    ME_TRY(exclusive_scan_u32_plain(ctx, wave_runs, wave_off, nw));  // (plain kernels: this build runs beside the other lane's full-chip kernels)
    unsigned int last_off = 0, last_runs = 0;
    ME_TRACE_POINT(ctx, "voxel_build: count_runs + scan queued");
    int h_err = 0;
    {
        MailGuard mg(ctx);  // (one synchronisation for the batch; destinations are locals of this frame)
        ME_TRY(mail_post(ctx, &last_off, wave_off + (nw - 1), 4));
        ME_TRY(mail_post(ctx, &last_runs, wave_runs + (nw - 1), 4));
        ME_TRY(mail_post(ctx, &h_err, d_err, 4));
        ME_TRY(mg.sync());
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err) return ctx->fail(ME_ERR_ARG, "voxel index out of range (|floor(p/voxel_size)| must be < 2^20)");
    ME_TRACE_POINT(ctx, "voxel_build: run counts read");
    const long long R = (long long) last_off + last_runs;  // run records
    // --- pass 1: (key, n, sum p) per run ---
    DevBuf &kbuf = ctx->tmp[1], &sbuf = ctx->tmp[2], &ibuf = ctx->tmp[3], &mbuf = ctx->tmp[4];  // (tmp[5]: sort/scan scratch)
    ME_CHECK(ctx, kbuf.ensure((size_t) R * 16));          // rec_key | sorted keys
    ME_CHECK(ctx, sbuf.ensure((size_t) R * 24));          // rec_sum
    ME_CHECK(ctx, ibuf.ensure((size_t) R * 16));          // iota | perm_r | rec_n | rec_vox
    ME_CHECK(ctx, mbuf.ensure((size_t) R * 48 + 64));     // rec_m2; before pass 2 also the head flags | positions
    unsigned long long *rec_key = kbuf.as<unsigned long long>(), *skey = rec_key + R;
    double *rec_sum = sbuf.as<double>(), *rec_m2 = mbuf.as<double>();
    unsigned int *iota = ibuf.as<unsigned int>(), *perm_r = iota + R, *rec_vox = perm_r + 2 * R;
    int *rec_n = reinterpret_cast<int *>(perm_r + R);
    unsigned int *flags = mbuf.as<unsigned int>(), *pos = flags + R;
    {
        TimerScope ts(ctx, "voxel");
        hipLaunchKernelGGL(k_vox_pass1, dim3(grid_for(n)), dim3(256), 0, ctx->stream, sp, n, voxel_size, c.slab, wave_off, rec_key,
                           iota, rec_n, rec_sum, d_err);
    }
    // radix sort is stable: the sorted order is preserved inside every voxel (fixed summation order)
#if ME_TUNE_VOX_MERGE_SORT
    ME_TRY(sort_pairs_merge_u64_u32(ctx, rec_key, skey, iota, perm_r, R));  // (stable, like the radix sort: same order, plain kernels)
#else
    ME_TRY(sort_pairs_u64_u32(ctx, rec_key, skey, iota, perm_r, R, 0, 63));
#endif
    hipLaunchKernelGGL(k_head_flags, dim3(grid_for(R)), dim3(256), 0, ctx->stream, skey, R, flags);
    ME_TRY(exclusive_scan_u32_plain(ctx, flags, pos, R));
    unsigned int last_pos = 0, last_flag = 0;
    unsigned long long last_key = 0;
    {
        MailGuard mg(ctx);  // (one synchronisation for the batch; destinations are locals of this frame)
        ME_TRY(mail_post(ctx, &last_pos, pos + (R - 1), 4));
        ME_TRY(mail_post(ctx, &last_flag, flags + (R - 1), 4));
        ME_TRY(mail_post(ctx, &last_key, skey + (R - 1), 8));
        ME_TRY(mg.sync());
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    long long V = (long long) last_pos + last_flag;
    ME_TRACE_POINT(ctx, "voxel_build: records sorted, voxel count read");
    const bool has_sentinel = c.slab.axis >= 0 && last_key == kVoxSentinel;
    if (has_sentinel) V -= 1;  // the halo points were keyed with the sentinel: their segment (the last one) is dropped
    DevBuf &seg = wbuf;        // wave_runs is dead (wave_off lives in the upper half)
    unsigned int *seg_start = wave_runs;
    if ((size_t) (V + 2) * 4 > (size_t) nw * 4) {  // (V + 1 <= R <= n, but not necessarily <= nw)
        ME_CHECK(ctx, c.vox_tmp.ensure((size_t) (V + 2) * 4));
        seg_start = c.vox_tmp.as<unsigned int>();
    }
    (void) seg;
    ME_CHECK(ctx, c.vox_key.ensure((size_t) (V + 1) * 8));
    ME_CHECK(ctx, c.vox_n.ensure((size_t) std::max<long long>(V, 1) * 4));
    ME_CHECK(ctx, c.vox_mu.ensure((size_t) std::max<long long>(V, 1) * 24));
    ME_CHECK(ctx, c.vox_sigma.ensure((size_t) std::max<long long>(V, 1) * 72));
    ME_CHECK(ctx, c.vox_entropy.ensure((size_t) std::max<long long>(V, 1) * 8));
    hipLaunchKernelGGL(k_seg_scatter, dim3(grid_for(R)), dim3(256), 0, ctx->stream, skey, flags, pos, R,
                       c.vox_key.as<unsigned long long>(), seg_start);
    if (!has_sentinel)  // (with a sentinel segment, entry V is its start, written by the scatter)
        hipLaunchKernelGGL(k_set_u32v, dim3(1), dim3(1), 0, ctx->stream, seg_start, V, (unsigned int) R);
    if (V > 0) {
        TimerScope ts(ctx, "voxel");
        const dim3 gv((unsigned int) ((V + 3) / 4));
        hipLaunchKernelGGL(k_vox_mean, gv, dim3(256), 0, ctx->stream, perm_r, seg_start, V, rec_n, rec_sum, rec_vox, c.vox_n.as<int>(),
                           c.vox_mu.as<double>());
        hipLaunchKernelGGL(k_vox_pass2, dim3(grid_for(n)), dim3(256), 0, ctx->stream, sp, n, voxel_size, c.slab, wave_off, rec_vox,
                           c.vox_mu.as<double>(), rec_m2, d_err);
        hipLaunchKernelGGL(k_vox_final, gv, dim3(256), 0, ctx->stream, perm_r, seg_start, V, rec_m2, c.vox_n.as<int>(), raw ? 1 : 0,
                           c.vox_sigma.as<double>(), c.vox_entropy.as<double>());
    }
    ME_CHECK(ctx, hipGetLastError());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.n_vox = V;
    c.vox_size = voxel_size;
    c.vox_raw = raw;
    c.vox_valid = true;
    return ME_OK;
}

int voxel_export(me_ctx *ctx, int slot, int32_t *keys, int32_t *npts, double *mu, double *sigma, double *entropy,
                 int64_t *n_voxels) {
    Cloud &c = ctx->cloud[slot];
    if (!n_voxels) return ctx->fail(ME_ERR_ARG, "n_voxels is NULL");
    const long long V = c.n_vox;
    const long long cap = *n_voxels;
    *n_voxels = V;
    if (!keys && !npts && !mu && !sigma && !entropy) return ME_OK;
    if (cap < V) return ctx->fail(ME_ERR_CAPACITY, "me_voxel_gaussians: capacity too small");
    if (keys) {
        ME_CHECK(ctx, ctx->tmp[0].ensure((size_t) V * 12));
        hipLaunchKernelGGL(k_unpack_keys, dim3(grid_for(V)), dim3(256), 0, ctx->stream, c.vox_key.as<unsigned long long>(), V,
                           ctx->tmp[0].as<int>());
        ME_TRY(copy_d2h(ctx, keys, ctx->tmp[0].p, (size_t) V * 12));
    }
    if (npts) ME_TRY(copy_d2h(ctx, npts, c.vox_n.p, (size_t) V * 4));
    if (mu) ME_TRY(copy_d2h(ctx, mu, c.vox_mu.p, (size_t) V * 24));
    if (sigma) ME_TRY(copy_d2h(ctx, sigma, c.vox_sigma.p, (size_t) V * 72));
    if (entropy) ME_TRY(copy_d2h(ctx, entropy, c.vox_entropy.p, (size_t) V * 8));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int awd_scs(me_ctx *ctx, double voxel_size, int min_pts, int scs_radius, double *rows, double *w_sorted, int64_t *n_rows,
            double *awd, double *scs, int64_t counts[3]) {
    if (scs_radius < 1 || scs_radius > 10) return ctx->fail(ME_ERR_ARG, "scs_radius must be in [1, 10]");
    Cloud &E = ctx->cloud[ME_SLOT_EST], &G = ctx->cloud[ME_SLOT_GT];
    // multi-GPU: both tables were merged from every rank's partials (me_voxel_merge_device) and are complete here
    const bool merged = E.vox_merged && G.vox_merged && E.vox_valid && G.vox_valid && E.vox_size == voxel_size && G.vox_size == voxel_size;
    if (!merged) {
        if (E.slab.axis >= 0 || G.slab.axis >= 0)
            return ctx->fail(ME_ERR_STATE, "me_awd_scs: in slab mode merge the partials of all ranks first (me_voxel_merge_device)");
        ME_TRY(voxel_build(ctx, ME_SLOT_GT, voxel_size, false));
        ME_TRY(voxel_build(ctx, ME_SLOT_EST, voxel_size, false));
    }
    const long long Ve = E.n_vox, Vg = G.n_vox;
    DevBuf &gi = ctx->tmp[0], &match = ctx->tmp[1], &active = ctx->tmp[2], &mpos = ctx->tmp[3];
    ME_CHECK(ctx, gi.ensure((size_t) Ve * 4));
    ME_CHECK(ctx, match.ensure((size_t) Ve * 4));
    ME_CHECK(ctx, active.ensure((size_t) Ve * 4));
    ME_CHECK(ctx, mpos.ensure((size_t) Ve * 4));
    ME_CHECK(ctx, ctx->red.ensure(256));
    long long *d_cnt = ctx->red.as<long long>();  // [0] active, [1] matched, [2..] scratch
    hipLaunchKernelGGL(k_join, dim3(grid_for(Ve)), dim3(256), 0, ctx->stream, E.vox_key.as<unsigned long long>(),
                       E.vox_n.as<int>(), Ve, G.vox_key.as<unsigned long long>(), G.vox_n.as<int>(), Vg, min_pts, gi.as<int>(),
                       match.as<unsigned int>(), active.as<unsigned int>());
    hipLaunchKernelGGL(k_count_u32, dim3(1), dim3(256), 0, ctx->stream, active.as<unsigned int>(), Ve, d_cnt);
    hipLaunchKernelGGL(k_count_u32, dim3(1), dim3(256), 0, ctx->stream, match.as<unsigned int>(), Ve, d_cnt + 1);
    ME_TRY(exclusive_scan_u32(ctx, match.as<unsigned int>(), mpos.as<unsigned int>(), Ve));
    long long h_cnt[2] = {0, 0};
    {
        MailGuard mg(ctx);  // (one synchronisation for the batch; destinations are locals of this frame)
        ME_TRY(mail_post(ctx, h_cnt, d_cnt, 16));
        ME_TRY(mg.sync());
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const long long n_active = h_cnt[0], M = h_cnt[1];
    if (counts) {  // "Update active/old/new voxel num" (voxel_calculator.cpp:170)
        counts[0] = n_active;
        counts[1] = Vg - n_active;
        counts[2] = Ve - n_active;
    }
    const long long cap = n_rows ? *n_rows : 0;
    if (n_rows) *n_rows = M;
    const double nan = std::nan("");
    if (M == 0) {  // mean of an empty vector / 0-0 division: NaN, as map_eval.cpp:324 and :387
        if (awd) *awd = nan;
        if (scs) *scs = nan;
        return ME_OK;
    }
    if ((rows || w_sorted) && cap < M) return ctx->fail(ME_ERR_CAPACITY, "me_awd_scs: capacity too small");
    DevBuf &mkey = ctx->tmp[4], &mw = ctx->tmp[5];
    DevBuf rows_d, ws_d;
    ME_CHECK(ctx, mkey.ensure((size_t) M * 8));
    // tmp[5] doubles as rocPRIM scratch: W must live in its own buffer
    DevBuf mw_own;
    ME_CHECK(ctx, mw_own.ensure((size_t) M * 8));
    (void) mw;
    if (rows) ME_CHECK(ctx, rows_d.ensure((size_t) M * 27 * 8));
    {
        TimerScope ts(ctx, "w2");
        hipLaunchKernelGGL(k_w2, dim3(grid_for(Ve)), dim3(256), 0, ctx->stream, match.as<unsigned int>(), mpos.as<unsigned int>(),
                           gi.as<int>(), Ve, E.vox_key.as<unsigned long long>(), E.vox_n.as<int>(), E.vox_mu.as<double>(),
                           E.vox_sigma.as<double>(), G.vox_n.as<int>(), G.vox_mu.as<double>(), G.vox_sigma.as<double>(),
                           voxel_size, mkey.as<unsigned long long>(), mw_own.as<double>(), rows ? rows_d.as<double>() : nullptr);
    }
    // AWD = mean W (map_eval.cpp:324)
    double *d_s = reinterpret_cast<double *>(d_cnt + 4);
    hipLaunchKernelGGL(k_sum_masked, dim3(1), dim3(256), 0, ctx->stream, mw_own.as<double>(), (const unsigned int *) nullptr, M,
                       d_s, d_cnt + 2);
    // SCS (map_eval.cpp:347-389)
    double *d_s2 = d_s + 1;
    ME_TRY(scs_device(ctx, mkey.as<unsigned long long>(), mw_own.as<double>(), M, scs_radius, d_s2, d_cnt + 3));
    double h_s[2] = {0, 0};
    long long h_c[2] = {0, 0};
    {
        MailGuard mg(ctx);  // (one synchronisation for the batch; destinations are locals of this frame)
        ME_TRY(mail_post(ctx, h_s, d_s, 16));
        ME_TRY(mail_post(ctx, h_c, d_cnt + 2, 16));
        ME_TRY(mg.sync());
    }
    if (rows) ME_TRY(copy_d2h(ctx, rows, rows_d.p, (size_t) M * 27 * 8));
    if (w_sorted) {
        ME_CHECK(ctx, ws_d.ensure((size_t) M * 8));
        ME_TRY(sort_keys_f64(ctx, mw_own.as<double>(), ws_d.as<double>(), M));  // std::sort(ws_distances) (:330)
        ME_TRY(copy_d2h(ctx, w_sorted, ws_d.p, (size_t) M * 8));
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    if (awd) *awd = h_s[0] / (double) h_c[0];
    if (scs) *scs = h_s[1] / (double) h_c[1];  // 0/0 -> NaN when no voxel has a neighbour (:387)
    return ME_OK;
}

int voxel_downsample(me_ctx *ctx, int slot, double voxel_size, long long *n_out) {
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    if (!(voxel_size > 0)) return ctx->fail(ME_ERR_ARG, "me_voxel_downsample: voxel_size must be > 0");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "me_voxel_downsample: cloud not uploaded");
    if (c.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_voxel_downsample: not available in slab mode (down-sample before sharding)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = c.n;
    DevBuf &keys_in = ctx->tmp[0], &iota = ctx->tmp[1], &keys = ctx->tmp[2], &perm = ctx->tmp[3], &flags = ctx->tmp[4];
    ME_CHECK(ctx, keys_in.ensure((size_t) n * 8));
    ME_CHECK(ctx, iota.ensure((size_t) n * 4));
    ME_CHECK(ctx, keys.ensure((size_t) n * 8));
    ME_CHECK(ctx, perm.ensure((size_t) n * 4));
    ME_CHECK(ctx, flags.ensure((size_t) n * 4));
    ME_CHECK(ctx, ctx->red.ensure(64));
    int *d_err = ctx->red.as<int>();
    ME_CHECK(ctx, hipMemsetAsync(d_err, 0, 4, ctx->stream));
    // voxel_min_bound = GetMinBound() - voxel_size / 2  [Open3D, upstream]
    const double mx = c.bbox_lo[0] - voxel_size * 0.5, my = c.bbox_lo[1] - voxel_size * 0.5, mz = c.bbox_lo[2] - voxel_size * 0.5;
    TimerScope ts(ctx, "downsample");
    hipLaunchKernelGGL(k_vds_keys, dim3(grid_for(n)), dim3(256), 0, ctx->stream, c.xyz.as<double>(), n, voxel_size, mx, my, mz,
                       keys_in.as<unsigned long long>(), iota.as<unsigned int>(), d_err);
    ME_TRY(sort_pairs_u64_u32(ctx, keys_in.as<unsigned long long>(), keys.as<unsigned long long>(), iota.as<unsigned int>(),
                              perm.as<unsigned int>(), n, 0, 63));
    DevBuf &pos = ctx->tmp[1];
    hipLaunchKernelGGL(k_head_flags, dim3(grid_for(n)), dim3(256), 0, ctx->stream, keys.as<unsigned long long>(), n,
                       flags.as<unsigned int>());
    ME_TRY(exclusive_scan_u32(ctx, flags.as<unsigned int>(), pos.as<unsigned int>(), n));
    unsigned int last_pos = 0, last_flag = 0;
    int h_err = 0;
    {
        MailGuard mg(ctx);  // (one synchronisation for the batch; destinations are locals of this frame)
        ME_TRY(mail_post(ctx, &last_pos, pos.as<unsigned int>() + (n - 1), 4));
        ME_TRY(mail_post(ctx, &last_flag, flags.as<unsigned int>() + (n - 1), 4));
        ME_TRY(mail_post(ctx, &h_err, d_err, 4));
        ME_TRY(mg.sync());
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err) return ctx->fail(ME_ERR_ARG, "voxel_size is too small for the cloud extent (more than 2^21 voxels per axis)");
    const long long V = (long long) last_pos + last_flag;
    DevBuf &seg_start = ctx->tmp[0], seg_key, out;
    ME_CHECK(ctx, seg_start.ensure((size_t) (V + 1) * 4));
    ME_CHECK(ctx, seg_key.ensure((size_t) V * 8));
    ME_CHECK(ctx, out.ensure((size_t) V * 24));
    hipLaunchKernelGGL(k_seg_scatter, dim3(grid_for(n)), dim3(256), 0, ctx->stream, keys.as<unsigned long long>(),
                       flags.as<unsigned int>(), pos.as<unsigned int>(), n, seg_key.as<unsigned long long>(),
                       seg_start.as<unsigned int>());
    hipLaunchKernelGGL(k_set_u32v, dim3(1), dim3(1), 0, ctx->stream, seg_start.as<unsigned int>(), V, (unsigned int) n);
    hipLaunchKernelGGL(k_vds_mean, dim3(grid_for(V)), dim3(256), 0, ctx->stream, c.xyz.as<double>(), perm.as<unsigned int>(),
                       seg_start.as<unsigned int>(), V, out.as<double>());
    if (!c.xyz.owned) ME_CHECK(ctx, c.xyz.ensure((size_t) V * 24));  // (a borrowed input buffer is the caller's: the result gets its own)
    ME_CHECK(ctx, hipMemcpyAsync(c.xyz.p, out.p, (size_t) V * 24, hipMemcpyDeviceToDevice, ctx->stream));
    if (c.have_normals) {  // Open3D averages the normals of a voxel as well (sum / count, not re-normalised)
        hipLaunchKernelGGL(k_vds_mean, dim3(grid_for(V)), dim3(256), 0, ctx->stream, c.normals.as<double>(), perm.as<unsigned int>(),
                           seg_start.as<unsigned int>(), V, out.as<double>());
        ME_CHECK(ctx, hipMemcpyAsync(c.normals.p, out.p, (size_t) V * 24, hipMemcpyDeviceToDevice, ctx->stream));
    }
    c.have_cov = false;
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    c.n = V;
    c.n_total = V;
    if (n_out) *n_out = V;
    return cloud_finish(ctx, slot);  // output order: ascending voxel index (Open3D: hash-map iteration order)
}

}  // namespace me
