// me_vox_rows.hpp — the per-row voxel run records of the one-pass voxel build (me_voxel.hip), shared with the index build's gather
// (me_index.hip), which emits them while it has the sorted points in registers.
#pragma once

#include <algorithm>
#include <cmath>

#include "me_internal.hpp"

namespace me {

constexpr int kKeyBias = 1 << 20;

__device__ __host__ __forceinline__ unsigned long long pack_key(int kx, int ky, int kz) {
    return ((unsigned long long) (unsigned int) (kx + kKeyBias) << 42) |
           ((unsigned long long) (unsigned int) (ky + kKeyBias) << 21) | (unsigned long long) (unsigned int) (kz + kKeyBias);
}
__device__ __host__ __forceinline__ void unpack_key(unsigned long long k, int &kx, int &ky, int &kz) {
    kx = (int) ((k >> 42) & 0x1fffff) - kKeyBias;
    ky = (int) ((k >> 21) & 0x1fffff) - kKeyBias;
    kz = (int) (k & 0x1fffff) - kKeyBias;
}


// sum over the lanes of the same (contiguous) segment, delivered to the segment's first lane
__device__ __forceinline__ double seg_sum_to_head(double v, int seg, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double ov = __shfl_down(v, o, 64);
        const int os = __shfl_down(seg, o, 64);
        if (lane + o < 64 && os == seg) v += ov;
    }
    return v;
}
// Sum over the WHOLE wave delivered to lane 0, on the vector unit (round 6): four DPP row shifts leave each row's sum in its first
// lane, three readlanes add the rows.  The segmented sums above cost three ds_bpermute per value and stage — 108 LDS-pipe operations
// per wavefront in k_vox_pass2, which made the voxel passes LDS-bound (0.5 ms per pass and 50 M points where the 1.6 GB they read
// take 0.33) — and a 3 m voxel holds thousands of consecutive sorted points: nearly every wavefront is ONE run.
__device__ __forceinline__ double wave_sum_to_lane0(double v) {
#define ME_ROW_SHL_ADD(N)                                                                                                    \
    {                                                                                                                        \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + (N), 0xF, 0xF, true);                       \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + (N), 0xF, 0xF, true);                       \
        v += __hiloint2double(hi, lo);                                                                                       \
    }
    ME_ROW_SHL_ADD(1)
    ME_ROW_SHL_ADD(2)
    ME_ROW_SHL_ADD(4)
    ME_ROW_SHL_ADD(8)
#undef ME_ROW_SHL_ADD
    auto row = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return ((row(0) + row(16)) + (row(32) + row(48)));  // (wave-uniform; lane 0 uses it)
}

__device__ __forceinline__ int seg_sum_to_head_i(int v, int seg, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ov = __shfl_down(v, o, 64);
        const int os = __shfl_down(seg, o, 64);
        if (lane + o < 64 && os == seg) v += ov;
    }
    return v;
}


constexpr unsigned long long kVoxEmptySlot = ~0ULL;
constexpr unsigned int kVoxRegions = 4096;  // overflow regions (a power of two)

__device__ __forceinline__ unsigned long long vox_compact_key(double x, double y, double z, const VoxPack &vp, const SlabView &slab,
                                                             int *__restrict__ err, double &cx, double &cy, double &cz) {
    cx = cy = cz = 0.0;
    if (!slab_owned(slab, x, y, z)) return vp.sentinel;
    const double fx = floor(x / vp.vs), fy = floor(y / vp.vs), fz = floor(z / vp.vs);  // getVoxelIndex (voxel_calculator.cpp:241-245)
    const double lim = (double) (kKeyBias - 16);
    if (!(fabs(fx) < lim && fabs(fy) < lim && fabs(fz) < lim)) {
        *err = 1;
        return 0;
    }
    cx = (fx + 0.5) * vp.vs;
    cy = (fy + 0.5) * vp.vs;
    cz = (fz + 0.5) * vp.vs;
    const unsigned long long ux = (unsigned long long) ((int) fx - vp.min_x), uy = (unsigned long long) ((int) fy - vp.min_y),
                             uz = (unsigned long long) ((int) fz - vp.min_z);
    return (((ux << vp.bits_y) | uy) << vp.bits_z) | uz;
}

constexpr int kVoxRec = 9;  // doubles per record: sum d (3), sum d d^T (xx, xy, xz, yy, yz, zz)

// the records of one row of 64 consecutive sorted points (lane = point)
__device__ __forceinline__ void vox_emit_row(bool valid, long long i, double x, double y, double z, const VoxPack &vp, const SlabView &slab,
                                             int lane, unsigned long long *__restrict__ rec_key, int *__restrict__ rec_n,
                                             double *__restrict__ rec_s, unsigned int *__restrict__ rec_count, unsigned int n_rows,
                                             unsigned int region_size, int *__restrict__ err) {
    double cx, cy, cz;
    const unsigned long long key = valid ? vox_compact_key(x, y, z, vp, slab, err, cx, cy, cz) : ~0ULL;
    const unsigned long long prev = __shfl_up(key, 1, 64);
    const bool head = valid && (lane == 0 || key != prev);
    const unsigned long long hm = __ballot(head);
    if (!hm) return;  // (a row past the end)
    const int run_local = __popcll(hm & ((2ULL << lane) - 1ULL)) - 1;
    const double dx = valid ? x - cx : 0.0, dy = valid ? y - cy : 0.0, dz = valid ? z - cz : 0.0;
    double v[kVoxRec] = {dx, dy, dz, dx * dx, dx * dy, dx * dz, dy * dy, dy * dz, dz * dz};
    int cnt;
    if (hm == 1ULL) {  // the row is one run: plain sums on the vector unit
        cnt = __popcll(__ballot(valid));
#pragma unroll
        for (int k = 0; k < kVoxRec; ++k) v[k] = wave_sum_to_lane0(v[k]);
    } else {
        const int seg = valid ? run_local : 64 + lane;
        cnt = seg_sum_to_head_i(valid ? 1 : 0, seg, lane);
#pragma unroll
        for (int k = 0; k < kVoxRec; ++k) v[k] = seg_sum_to_head(v[k], seg, lane);
    }
    // record slots: two per row; the further runs of a row go to one of kVoxRegions regions behind the rows' slots (the region of
    // row r is r mod kVoxRegions: neighbouring rows — which cross the same voxel faces — spread over the regions), reserved with
    // one atomic per such row on the REGION's counter: on one counter for the whole cloud, 10^5 - 10^6 atomics went through one L2
    // channel (0.5 - 8 ms).  A region that overflows is counted, not stored: the host sees it and takes the three-pass build.
    unsigned int base = 0;
    const int extra = __popcll(hm) - 2;  // runs beyond the row's two slots
    const unsigned int region = (unsigned int) (i >> 6) & (kVoxRegions - 1);
    if (extra > 0) {
        if (lane == 0) base = atomicAdd(rec_count + region, (unsigned int) extra);
        base = (unsigned int) __builtin_amdgcn_readfirstlane((int) base);
    }
    if (head) {
        const bool fits = run_local < 2 || base + (unsigned int) (run_local - 2) < region_size;
        const unsigned int r = run_local < 2 ? 2u * (unsigned int) (i >> 6) + (unsigned int) run_local
                                             : 2u * n_rows + region * region_size + base + (unsigned int) (run_local - 2);
        if (fits) {
            rec_key[r] = (key << vp.pos_bits) | ((unsigned long long) (i >> 6) << 6) | (unsigned long long) run_local;
            rec_n[r] = cnt;
#pragma unroll
            for (int k = 0; k < kVoxRec; ++k) rec_s[(long long) kVoxRec * r + k] = v[k];
        }
    }
}


// ---- host side ----
inline int vox_bits_for(long long range) {  // bits that hold 0 .. range
    int b = 1;
    while ((1LL << b) <= range) ++b;
    return b;
}

// the compact sort key of the one-pass build for this cloud and voxel size; false: it does not fit 64 bits (three-pass build)
inline bool vox_make_pack(const Cloud &c, double vs, long long n, VoxPack &vp) {
    long long lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
        const double a = std::floor(c.bbox_lo[d] / vs), b = std::floor(c.bbox_hi[d] / vs);  // floor(x / vs) is monotone: the points' indices lie between
        if (!(std::fabs(a) < (double) (kKeyBias - 16) && std::fabs(b) < (double) (kKeyBias - 16))) return false;  // (the three-pass build reports the range error)
        lo[d] = (long long) a;
        hi[d] = (long long) b;
    }
    const int bx = vox_bits_for(hi[0] - lo[0]), by = vox_bits_for(hi[1] - lo[1]), bz = vox_bits_for(hi[2] - lo[2]);
    const int row_bits = vox_bits_for(std::max<long long>(1, (n - 1) >> 6));
    vp.vs = vs;
    vp.min_x = (int) lo[0];
    vp.min_y = (int) lo[1];
    vp.min_z = (int) lo[2];
    vp.bits_y = by;
    vp.bits_z = bz;
    vp.pos_bits = row_bits + 6;
    vp.sentinel = 1ULL << (bx + by + bz);  // one above every real compact key
    return bx + by + bz + 1 + vp.pos_bits <= 63;  // (bit 63 stays clear: kVoxEmptySlot is no record's key)
}
// record slots of a cloud of n points: two per row of 64, then kVoxRegions regions for the further runs — together as many again
// (a row crosses a voxel face, or holds an outlier of another voxel, more often than one would think: 10 - 70 % of the rows
// of the bench's clouds have three or more runs)
inline long long vox_region_size(long long n) { return (2 * ((n + 63) / 64) + kVoxRegions - 1) / kVoxRegions + 16; }
inline long long vox_record_capacity(long long n) { return 2 * ((n + 63) / 64) + (long long) kVoxRegions * vox_region_size(n); }
constexpr size_t kVoxCounterBytes = (size_t) (2 + kVoxRegions) * 4;  // [range error, max region fill (k_vox_region_max)] + the regions' counters

}  // namespace me
