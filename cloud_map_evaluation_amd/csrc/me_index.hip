// me_index.hip — cloud upload + spatial index build.
//
// Replaces every KDTreeFlann::SetGeometry of the reference (map_eval.cpp:1214,1227,1401-1402,1449,1551,1619 —
// 5-7 single-threaded KD-tree builds per run) with ONE index per cloud:
//   1. optional rigid/homogeneous transform (map_eval.cpp:1206) + bbox reduction,
//   2. 63-bit Morton codes on a grid nested in the radius-search cell (cell = code >> 3*shift),
//   3. radix sort (rocPRIM) and gather into 32-byte SPoint records -> every later pass streams coalesced,
//   4. the tables of occupied cells (radius grid, 1-NN grid) + open-addressing hashes (cell Morton code -> cell),
//   5. a sparse octree over the Morton prefixes above the 1-NN cells (tight fp32 boxes rounded outward).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "me_internal.hpp"
#include "me_vox_rows.hpp"

namespace me {

// ------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------
struct Mat4 {
    double m[16];
};

__global__ void k_transform(double *__restrict__ xyz, long long n, Mat4 T) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    // Open3D PointCloud::Transform [upstream]: h = T*(x,y,z,1), p = h.xyz / h.w — column-major accumulation
    // order ((c0*x + c1*y) + c2*z) + c3, no FMA (must match the CPU path bit for bit).
    double h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = ((T.m[4 * r] * x + T.m[4 * r + 1] * y) + T.m[4 * r + 2] * z) + T.m[4 * r + 3];
    xyz[3 * i] = h[0] / h[3];
    xyz[3 * i + 1] = h[1] / h[3];
    xyz[3 * i + 2] = h[2] / h[3];
}

__global__ void __launch_bounds__(256) k_bbox(const double *__restrict__ xyz, long long n, double *__restrict__ part) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double v = xyz[3 * i + d];
            lo[d] = fmin(lo[d], v);
            hi[d] = fmax(hi[d], v);
        }
    }
    __shared__ double sm[4][6];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fmin(lo[d], __shfl_down(lo[d], o, 64));
            hi[d] = fmax(hi[d], __shfl_down(hi[d], o, 64));
        }
        if (lane == 0) {
            sm[w][d] = lo[d];
            sm[w][3 + d] = hi[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        part[6 * blockIdx.x + d] = fmin(fmin(sm[0][d], sm[1][d]), fmin(sm[2][d], sm[3][d]));
        part[6 * blockIdx.x + 3 + d] = fmax(fmax(sm[0][3 + d], sm[1][3 + d]), fmax(sm[2][3 + d], sm[3][3 + d]));
    }
}

// slab mode: keep the points inside [reg_lo, reg_hi) along the slab axis (owned + halo)
// Open3D PointCloud::Transform on one point, the arithmetic of k_transform
__device__ __forceinline__ void transform_point(const Mat4 &T, double x, double y, double z, double *o) {
    double h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = ((T.m[4 * r] * x + T.m[4 * r + 1] * y) + T.m[4 * r + 2] * z) + T.m[4 * r + 3];
    o[0] = h[0] / h[3];
    o[1] = h[1] / h[3];
    o[2] = h[2] / h[3];
}

// slab filter straight from the caller's buffer: flag pass and compaction both apply the optional transform on the fly
// (no staged copy of the whole cloud; every rank of a multi-GPU job runs this over ALL points, so it is kept lean)
__global__ void k_slab_flags(const double *__restrict__ xyz, long long n, int has_T, Mat4 T, SlabView s,
                             unsigned int *__restrict__ flags) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (has_T) transform_point(T, p[0], p[1], p[2], p);
    const double v = p[s.axis];
    flags[i] = (v >= s.reg_lo && v < s.reg_hi) ? 1u : 0u;
}
__global__ void k_slab_compact(const double *__restrict__ xyz, long long n, int has_T, Mat4 T,
                               const unsigned int *__restrict__ flags, const unsigned int *__restrict__ pos,
                               double *__restrict__ out, int *__restrict__ orig) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const long long o = pos[i];
    orig[o] = (int) i;  // per-point outputs in slab mode are reported against the uploaded array (me_slab_points)
    double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (has_T) transform_point(T, p[0], p[1], p[2], p);
    out[3 * o] = p[0];
    out[3 * o + 1] = p[1];
    out[3 * o + 2] = p[2];
}

// upload from a device buffer: copy + optional transform + bounding-box partials in ONE pass over the cloud (three
// passes otherwise: the copy, k_transform, k_bbox).  Same arithmetic as k_transform / k_bbox; part = [blocks][6].
__global__ void __launch_bounds__(256)
k_ingest(const double *__restrict__ in, long long n, int has_T, Mat4 T, double *__restrict__ out, double *__restrict__ part) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        double p[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]};
        if (has_T) transform_point(T, p[0], p[1], p[2], p);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            out[3 * i + d] = p[d];
            lo[d] = fmin(lo[d], p[d]);
            hi[d] = fmax(hi[d], p[d]);
        }
    }
    __shared__ double sm[4][6];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fmin(lo[d], __shfl_down(lo[d], o, 64));
            hi[d] = fmax(hi[d], __shfl_down(hi[d], o, 64));
        }
        if (lane == 0) {
            sm[w][d] = lo[d];
            sm[w][3 + d] = hi[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        part[6 * blockIdx.x + d] = fmin(fmin(sm[0][d], sm[1][d]), fmin(sm[2][d], sm[3][d]));
        part[6 * blockIdx.x + 3 + d] = fmax(fmax(sm[0][3 + d], sm[1][3 + d]), fmax(sm[2][3 + d], sm[3][3 + d]));
    }
}

// Hilbert index of a cell (x, y, z), b bits per axis, as a 3b-bit key whose every 3-bit group is one level of the curve
// (Skilling, "Programming the Hilbert curve", AIP Conf. Proc. 707, 2004: the transpose form, interleaved).  Like the Morton
// code it is hierarchical — the top 3k bits depend on the top k bits of the coordinates only, so every aligned 2^j block of
// cells is one contiguous run of the sorted array, which is all the cell tables and the octree ask of the order — but
// consecutive cells along it are always face neighbours: 64 consecutive points never jump across the scene the way they
// do at the seams of the Z curve.
__device__ __forceinline__ unsigned long long hilbert_key(unsigned int x, unsigned int y, unsigned int z, int b) {
    const unsigned int M = 1u << (b - 1);
    for (unsigned int Q = M; Q > 1; Q >>= 1) {
        const unsigned int P = Q - 1;
        if (x & Q) x ^= P;
        if (y & Q) {
            x ^= P;
        } else {
            const unsigned int t = (x ^ y) & P;
            x ^= t;
            y ^= t;
        }
        if (z & Q) {
            x ^= P;
        } else {
            const unsigned int t = (x ^ z) & P;
            x ^= t;
            z ^= t;
        }
    }
    y ^= x;
    z ^= y;
    unsigned int t = 0;
    for (unsigned int Q = M; Q > 1; Q >>= 1)
        if (z & Q) t ^= Q - 1;
    x ^= t;
    y ^= t;
    z ^= t;
    return (spread21((unsigned long long) x) << 2) | (spread21((unsigned long long) y) << 1) | spread21((unsigned long long) z);
}

// fine-lattice coordinates of a point (the Morton code interleaves them)
__device__ __forceinline__ void fine_cell(const double *__restrict__ xyz, long long i, double ox, double oy, double oz,
                                          double fine_h, unsigned int &cx, unsigned int &cy, unsigned int &cz) {
    const double lim = 2097151.0;
    // division (not multiply-by-reciprocal): (p-o)/(h*2^-s) == ((p-o)/h)*2^s exactly, so the radius cell
    // floor((p-o)/h) is exactly the top bits of the fine coordinate.
    cx = (unsigned int) fmin(fmax(fine_coord(xyz[3 * i], ox, fine_h), 0.0), lim);
    cy = (unsigned int) fmin(fmax(fine_coord(xyz[3 * i + 1], oy, fine_h), 0.0), lim);
    cz = (unsigned int) fmin(fmax(fine_coord(xyz[3 * i + 2], oz, fine_h), 0.0), lim);
}

// sort keys: the Hilbert index of the point's cell at level `min_level` (the finest level the order matters at), placed at
// bits [3 min_level, 63) — or the Morton code itself (hilbert == 0)
// pack_bits > 0 (round 4): key and point index travel in ONE 64-bit word, (key >> 3 min_level) << pack_bits | i, and the sort is a
// keys-only sort of the bits above pack_bits — 16 bytes per point and radix pass instead of 24 (8-byte key + 4-byte value, read and
// written); the index in the low bits also breaks ties the way the stable pair sort did: the permutation is the same.
__global__ void k_morton(const double *__restrict__ xyz, long long n, double ox, double oy, double oz, double fine_h,
                         int min_level, int hilbert, int pack_bits, unsigned long long *__restrict__ keys, unsigned int *__restrict__ iota) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned int cx, cy, cz;
    fine_cell(xyz, i, ox, oy, oz, fine_h, cx, cy, cz);
    unsigned long long key;
    if (hilbert)
        key = hilbert_key(cx >> min_level, cy >> min_level, cz >> min_level, kMortonBits - min_level) << (3 * min_level);
    else
        key = spread21((unsigned long long) cx) | (spread21((unsigned long long) cy) << 1) | (spread21((unsigned long long) cz) << 2);
    if (pack_bits > 0) {
        keys[i] = ((key >> (3 * min_level)) << pack_bits) | (unsigned long long) i;
    } else {
        keys[i] = key;
        iota[i] = (unsigned int) i;
    }
}

// sorted points, and their Morton codes (the cell keys of every level; recomputed here rather than carried through the sort)
// (packed != nullptr: the sorted (key, index) words of the keys-only sort, the index in the low bits under idx_mask)
constexpr int kGatherPer = 4;  // sorted points per thread (round 6): the four index loads, then the twelve coordinate loads of a thread are in flight together
constexpr int kGatherPerVox = 2;  // ... two when the gather also emits the voxel run records: with four it needs 88 vector registers, and a block
// that needs more registers than a retiring k_mme3 block frees (64 per lane) waits for two of them to retire on the same SIMDs — the
// ground truth's gather took 14 ms beside the map's MME (profiles/EXPERIMENTS.md "Round 6")
// VOX (round 6): the gather also emits the run records of the one-pass voxel build (me_vox_rows.hpp) for the voxel size the context
// carries as a hint (me_run_suite_from): it has the sorted points in registers and waits on random reads, the records' arithmetic
// costs it little — and the voxel build that follows has no pass over the cloud left.
template <bool VOX, int PER>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_gather(const double *__restrict__ xyz, const unsigned int *__restrict__ perm, const unsigned long long *__restrict__ packed,
         unsigned long long idx_mask, long long n, double ox, double oy, double oz, double fine_h,
         SPoint *__restrict__ sp, unsigned long long *__restrict__ codes, VoxPack vp, SlabView slab,
         unsigned long long *__restrict__ rec_key, int *__restrict__ rec_n, double *__restrict__ rec_s, unsigned int *__restrict__ rec_cnt,
         unsigned int n_rows, unsigned int rec_cap) {
    // (a block covers 256 x PER consecutive sorted points, a thread the points base + k * 256 + threadIdx.x: every store is coalesced)
    const long long base = (long long) blockIdx.x * (256 * PER) + threadIdx.x;
    unsigned int s[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const long long i = base + 256 * k;
        s[k] = i < n ? (packed ? (unsigned int) (packed[i] & idx_mask) : perm[i]) : 0u;
    }
    double x[PER], y[PER], z[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        x[k] = xyz[3 * (long long) s[k]];
        y[k] = xyz[3 * (long long) s[k] + 1];
        z[k] = xyz[3 * (long long) s[k] + 2];
    }
    const double lim = 2097151.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const long long i = base + 256 * k;
        if (i >= n) continue;
        SPoint p;
        p.x = x[k];
        p.y = y[k];
        p.z = z[k];
        p.idx = (long long) s[k];
        sp[i] = p;
        // (fine_cell's arithmetic on the values already in registers: division, not reciprocal multiply)
        const unsigned int cx = (unsigned int) fmin(fmax(fine_coord(x[k], ox, fine_h), 0.0), lim);
        const unsigned int cy = (unsigned int) fmin(fmax(fine_coord(y[k], oy, fine_h), 0.0), lim);
        const unsigned int cz = (unsigned int) fmin(fmax(fine_coord(z[k], oz, fine_h), 0.0), lim);
        codes[i] = spread21((unsigned long long) cx) | (spread21((unsigned long long) cy) << 1) | (spread21((unsigned long long) cz) << 2);
    }
    if (VOX) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {  // (every lane of the row takes part: shuffles and ballots inside)
            const long long i = base + 256 * k;
            vox_emit_row(i < n, i, x[k], y[k], z[k], vp, slab, threadIdx.x & 63, rec_key, rec_n, rec_s, rec_cnt + 2, n_rows, rec_cap,
                         reinterpret_cast<int *>(rec_cnt));
        }
    }
}

// octree level 0: one node per occupied 1-NN-grid cell (a contiguous run of sorted points).  Eight lanes per cell (round 4): they
// read the run eight points — one 256-byte burst — at a time and min / max their outward-rounded FP32 bounds with three DPP
// stages.  One lane per cell walked its ~25 points alone: 64 strided 32-byte reads per load instruction (0.63 ms per cloud).
__global__ void __launch_bounds__(256)
k_oct_leaves(const SPoint *__restrict__ sp, const unsigned int *__restrict__ cell_start, long long n_cells,
             long long n_points, ONode *__restrict__ nodes, unsigned int *__restrict__ pbegin) {
    const long long c = ((long long) blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    long long b = 0, e = 0;
    if (c < n_cells) {
        b = cell_start[c];
        e = cell_start[c + 1];
    }
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long j = b + sub; __ballot(j < e); j += 8) {
        if (j < e) {
            const SPoint p = sp[j];
            lo[0] = fmin(lo[0], p.x); hi[0] = fmax(hi[0], p.x);
            lo[1] = fmin(lo[1], p.y); hi[1] = fmax(hi[1], p.y);
            lo[2] = fmin(lo[2], p.z); hi[2] = fmax(hi[2], p.z);
        }
    }
    // outward rounding keeps the fp32 box a superset of the fp64 one (rounding is monotone: the minimum of the rounded values
    // is the rounded minimum)
    float flo[3], fhi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        flo[d] = __double2float_rd(lo[d]);
        fhi[d] = __double2float_ru(hi[d]);
        flo[d] = fminf(flo[d], __int_as_float(octet_partner_i<0>(__float_as_int(flo[d]))));
        fhi[d] = fmaxf(fhi[d], __int_as_float(octet_partner_i<0>(__float_as_int(fhi[d]))));
        flo[d] = fminf(flo[d], __int_as_float(octet_partner_i<1>(__float_as_int(flo[d]))));
        fhi[d] = fmaxf(fhi[d], __int_as_float(octet_partner_i<1>(__float_as_int(fhi[d]))));
        flo[d] = fminf(flo[d], __int_as_float(octet_partner_i<2>(__float_as_int(flo[d]))));
        fhi[d] = fmaxf(fhi[d], __int_as_float(octet_partner_i<2>(__float_as_int(fhi[d]))));
    }
    if (sub != 0 || c > n_cells) return;
    ONode nd;
    if (c == n_cells) {  // terminator: holds the end of the last run
        for (int d = 0; d < 3; ++d) nd.lo[d] = nd.hi[d] = 0.0f;
        nd.begin = (unsigned int) n_points;
        nd.parent = 0;
        nodes[c] = nd;
        pbegin[c] = (unsigned int) n_points;
        return;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        nd.lo[d] = flo[d];
        nd.hi[d] = fhi[d];
    }
    nd.begin = (unsigned int) b;
    nd.parent = 0;
    nodes[c] = nd;
    pbegin[c] = (unsigned int) b;
}

// octree level l+1 from level l: parent p owns the children [begin[p], begin[p+1]) (<= 8, contiguous)
__global__ void k_oct_up(ONode *__restrict__ child, long long n_child, const unsigned int *__restrict__ begin,
                         long long n_parent, ONode *__restrict__ parent, const unsigned int *__restrict__ child_pbegin,
                         unsigned int *__restrict__ parent_pbegin) {
    const long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_parent) return;
    ONode nd;
    if (p == n_parent) {
        for (int d = 0; d < 3; ++d) nd.lo[d] = nd.hi[d] = 0.0f;
        nd.begin = (unsigned int) n_child;
        nd.parent = 0;
        parent[p] = nd;
        parent_pbegin[p] = child_pbegin[n_child];  // (the child level's terminator: the point count)
        return;
    }
    const long long b = begin[p], e = (p + 1 < n_parent) ? (long long) begin[p + 1] : n_child;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long c = b; c < e; ++c) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], child[c].lo[d]);
            hi[d] = fmaxf(hi[d], child[c].hi[d]);
        }
        child[c].parent = (unsigned int) p;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        nd.lo[d] = lo[d];
        nd.hi[d] = hi[d];
    }
    nd.begin = (unsigned int) b;
    nd.parent = 0;
    parent[p] = nd;
    parent_pbegin[p] = child_pbegin[b];
}

// The TOP of the octree in one launch (round 4): every level above `l0` has at most kOctTopMax nodes; one 1024-thread block builds
// them all, level after level — what took a scan, k_cell_scatter and k_oct_up launch PER LEVEL (about ten levels of a few thousand
// nodes down to one: ~40 launch-sized kernels per cloud, nothing else).  Same node records as k_oct_up writes.
constexpr int kOctTopMax = 16384;
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_oct_top(ONode *__restrict__ nodes, unsigned int *__restrict__ pbeg, OctView v, int l0, const unsigned long long *__restrict__ codes0,
          unsigned long long *__restrict__ ca, unsigned long long *__restrict__ cb, unsigned int *__restrict__ begin) {
    constexpr int NW = BLOCK / 64;
    __shared__ unsigned int s_cnt[NW];
    const int t = threadIdx.x;
    const unsigned long long *cur = codes0;
    for (int l = l0; l + 1 < v.n_levels; ++l) {
        const long long nc = v.count[l], np = v.count[l + 1];
        unsigned long long *nxt = ((l - l0) & 1) ? cb : ca;
        ONode *child = nodes + v.off[l], *parent = nodes + v.off[l + 1];
        const unsigned int *cpb = pbeg + v.off[l];
        unsigned int *ppb = pbeg + v.off[l + 1];
        // parents start where the 3-bit-shorter prefix changes: rank the flags in order (ballot inside a wave, the wave totals
        // through LDS), BLOCK children per sweep, coalesced
        unsigned int running = 0;
        for (long long base = 0; base < nc; base += BLOCK) {
            const long long i = base + t;
            const bool flag = i < nc && (i == 0 || (cur[i] >> 3) != (cur[i - 1] >> 3));
            const unsigned long long m = __ballot(flag);
            const int lane = t & 63, wv = t >> 6;
            if (lane == 0) s_cnt[wv] = (unsigned int) __popcll(m);
            __syncthreads();
            unsigned int before = 0, total = 0;
            for (int w = 0; w < NW; ++w) {
                const unsigned int c = s_cnt[w];
                if (w < wv) before += c;
                total += c;
            }
            if (flag) {
                const unsigned int pos = running + before + (unsigned int) __popcll(m & ((1ULL << lane) - 1ULL));
                nxt[pos] = cur[i] >> 3;
                begin[pos] = (unsigned int) i;
            }
            running += total;
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        // parent records (k_oct_up)
        for (long long p = t; p <= np; p += BLOCK) {
            ONode nd;
            if (p == np) {
                for (int d = 0; d < 3; ++d) nd.lo[d] = nd.hi[d] = 0.0f;
                nd.begin = (unsigned int) nc;
                nd.parent = 0;
                parent[p] = nd;
                ppb[p] = cpb[nc];
                continue;
            }
            const long long b = begin[p], e = (p + 1 < np) ? (long long) begin[p + 1] : nc;
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (long long c = b; c < e; ++c) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    lo[d] = fminf(lo[d], child[c].lo[d]);
                    hi[d] = fmaxf(hi[d], child[c].hi[d]);
                }
                child[c].parent = (unsigned int) p;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                nd.lo[d] = lo[d];
                nd.hi[d] = hi[d];
            }
            nd.begin = (unsigned int) b;
            nd.parent = 0;
            parent[p] = nd;
            ppb[p] = cpb[b];
        }
        __threadfence_block();
        __syncthreads();
        cur = nxt;
    }
}

// the first point of every cell (code >> shift3 differs from the predecessor's) writes the cell's code and start; pos[i] =
// cell starts before i (cell_start_ranks)
__global__ void k_cell_scatter(const unsigned long long *__restrict__ codes, const unsigned int *__restrict__ pos, long long n,
                               int shift3, unsigned long long *__restrict__ cell_code, unsigned int *__restrict__ cell_start) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long c = codes[i] >> shift3;
    if (i == 0 || c != (codes[i - 1] >> shift3)) {
        cell_code[pos[i]] = c;
        cell_start[pos[i]] = (unsigned int) i;
    }
}

// (the level histogram — hist[k] += 1 when codes[i-1] and codes[i] first differ at level k (bits 3k..3k+2); #occupied cells at level k =
//  1 + sum_{j >= k} hist[j] — comes from the per-block rows of k_level_hist_rows below)
// ---- cell tables in two passes over the codes (round 5) ----------------------------------------------------------------------------
// Through round 4 a cell table cost a rank scan of per-point start flags (rocPRIM: codes read, 4 bytes per point written), a scatter
// that read codes and ranks again, and the hash inserts as a third kernel over the cell codes — after the level histogram had
// already streamed the same codes once.  But the histogram knows everything the scan computed: a point starts a cell of level L
// exactly when the highest level at which its code differs from its predecessor's is >= L, so a block's row of the level histogram
// gives the number of cells of ANY level that start inside the block.  Now:
//   k_level_hist_rows    per block of kCellChunk sorted points: its row of the level histogram — the one pass that was the histogram's;
//   k_block_counts       one thread per block: its cell count for a level (and, the first time, the rows added up), then a scan
//                        over the 24 k counts (a first version did both in ONE 1024-thread block: 0.27 ms alone — 2.3 MB through one
//                        CU, strided — and 2.6 ms when the other lane held the chip: a single block queues behind everything);
//   k_cell_fill          the chunks again: start flags ranked in order with ballots (4 wave counts through LDS per 256 points), the
//                        cell's code, its start and its hash entry written by the lane that found it.
// Per table 8 bytes per point read instead of 8 read + 4 written (scan) and 12 read (scatter), two launches fewer, the same table entry
// for entry (cells in sorted order).  Short blocks (2048 points each), no lookback and no persistent grid: a first version with 2048
// long-running blocks was 0.9 ms per step faster by its own timer and 0.5 ms SLOWER in the two-lane step — it held every wave slot of
// the chip for its whole duration and the other lane's kernels queued behind it (the round-4 lesson, again).
constexpr int kCellChunk = 2048;   // sorted points per block
constexpr int kHistLevels = 24;    // histogram rows hold levels 0 .. 23 (kMortonBits = 21)
static_assert(kHistLevels > kMortonBits, "a histogram row holds every Morton level");
__global__ void __launch_bounds__(256)
k_level_hist_rows(const unsigned long long *__restrict__ codes, long long n, unsigned int *__restrict__ block_hist) {
    __shared__ unsigned int sh[32];
    if (threadIdx.x < 32) sh[threadIdx.x] = 0;
    __syncthreads();
    const long long i0 = (long long) blockIdx.x * kCellChunk, i1 = i0 + kCellChunk < n ? i0 + kCellChunk : n;
    // (round 6: the chunk's eight rows of 256 codes are loaded before the first is looked at — eight loads in flight per lane instead of
    // eight dependent load -> ballot rounds)
    unsigned long long x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long i = i0 + 256 * k + threadIdx.x;
        x[k] = (i < i1 && i > 0) ? (codes[i] ^ codes[i - 1]) : 0ULL;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int lv = x[k] ? (63 - __clzll((long long) x[k])) / 3 : -1;
        unsigned long long todo = __ballot(lv >= 0);
        while (todo) {  // (wave-aggregated: neighbours differ at two or three distinct levels per wavefront)
            const int l0 = __builtin_amdgcn_readlane(lv, __ffsll((long long) todo) - 1);
            const unsigned long long same = __ballot(lv == l0);
            if ((threadIdx.x & 63) == 0) atomicAdd(&sh[l0], (unsigned int) __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    // (level-major: the rows of all blocks for one level lie together — k_block_counts reads them coalesced)
    if (threadIdx.x < kHistLevels) block_hist[(long long) threadIdx.x * gridDim.x + blockIdx.x] = sh[threadIdx.x];
}

// cnt[b] = cells of level `level` that start in block b (cnt[nb] = 0: the scan's last entry is the total); with HIST the rows are also
// added up into hist[0 .. kHistLevels) (zeroed by the caller).  One thread per block of the first pass.
template <bool HIST>
__global__ void __launch_bounds__(256)
k_block_counts(const unsigned int *__restrict__ block_hist, int nb, int level, unsigned int *__restrict__ cnt, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int sh[kHistLevels];
    if (HIST) {
        if (threadIdx.x < kHistLevels) sh[threadIdx.x] = 0;
        __syncthreads();
    }
    const int b = blockIdx.x * 256 + threadIdx.x;
    unsigned int c = b == 0 ? 1u : 0u;  // the first point starts a cell of every level
#pragma unroll
    for (int k = 0; k < kHistLevels; ++k) {
        if (!HIST && k < level) continue;
        const unsigned int r = b < nb ? block_hist[(long long) k * nb + b] : 0u;
        if (k >= level) c += r;
        if (HIST) {
            unsigned int w = r;  // (a row entry is <= 2048, a block of 256 of them fits 32 bits with room)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) w += __shfl_xor(w, d);
            if ((threadIdx.x & 63) == 0 && w) atomicAdd(&sh[k], w);
        }
    }
    if (b <= nb) cnt[b] = b < nb ? c : 0u;
    if (HIST) {
        __syncthreads();
        if (threadIdx.x < kHistLevels && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long) sh[threadIdx.x]);
    }
}

// (round 6) A wavefront owns 512 CONSECUTIVE points of the block's chunk, eight per lane, and loads all of them (and their predecessors)
// before it looks at any: the first version walked the chunk 256 points at a time with two block barriers per step — eight dependent
// load -> ballot -> LDS -> barrier rounds per block, 0.50 ms per 50 M points for 0.4 GB read.  One barrier per block now.
__global__ void __launch_bounds__(256)
k_cell_fill(const unsigned long long *__restrict__ codes, long long n, int shift3, const unsigned int *__restrict__ block_off,
            long long n_cells, unsigned long long *__restrict__ cell_code, unsigned int *__restrict__ cell_start,
            unsigned long long *__restrict__ hkeys, unsigned int *__restrict__ hvals, unsigned int hmask) {
    static_assert(kCellChunk == 2048, "four wavefronts x eight rows of 64 points");
    __shared__ unsigned int s_w[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long i0 = (long long) blockIdx.x * kCellChunk, i1 = i0 + kCellChunk < n ? i0 + kCellChunk : n;
    const long long w0 = i0 + (long long) wv * 512;
    unsigned long long c[8], cp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long i = w0 + 64 * k + lane;
        c[k] = i < i1 ? codes[i] : 0ULL;
        cp[k] = (i < i1 && i > 0) ? codes[i - 1] : 0ULL;
    }
    unsigned long long m[8];
    unsigned int wave_total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long i = w0 + 64 * k + lane;
        c[k] >>= shift3;
        const bool start = i < i1 && (i == 0 || c[k] != (cp[k] >> shift3));
        m[k] = __ballot(start);
        wave_total += (unsigned int) __popcll(m[k]);
    }
    if (lane == 0) s_w[wv] = wave_total;
    __syncthreads();
    unsigned int pos0 = block_off[blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wv) pos0 += s_w[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((m[k] >> lane) & 1ULL) {
            const unsigned int pos = pos0 + (unsigned int) __popcll(m[k] & ((1ULL << lane) - 1ULL));
            cell_code[pos] = c[k];
            cell_start[pos] = (unsigned int) (w0 + 64 * k + lane);
            unsigned int sl = (unsigned int) hash_u64(c[k]) & hmask;  // (open addressing, linear probe: first free slot)
            for (;;) {
                const unsigned long long prev = atomicCAS(&hkeys[sl], kEmptyKey, c[k]);
                if (prev == kEmptyKey || prev == c[k]) {
                    hvals[sl] = pos;
                    break;
                }
                sl = (sl + 1) & hmask;
            }
        }
        pos0 += (unsigned int) __popcll(m[k]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cell_start[n_cells] = (unsigned int) n;
}

static inline unsigned int grid_for(long long n, int block = 256) { return (unsigned int) ((n + block - 1) / block); }

// occupied cells of Morton level `shift` (unique code >> 3*shift, run starts) + open-addressing hash:
// a cell table of the sorted points from the blocks' histogram rows (see the kernels above); `offsets_ready`: block_off already holds this
// level's scan (block_off is followed by the nblk + 2 counts the scan reads: the layout of cloud_build_index)
static int build_grid_table_counted(me_ctx *ctx, Cloud &c, int shift, GridTable &t, GridView &g, int nblk, const unsigned int *block_hist,
                                    unsigned int *block_off, bool offsets_ready) {
    const long long n = c.n;
    if (!offsets_ready) {
        unsigned int *cnt = block_off + (nblk + 2);
        hipLaunchKernelGGL(k_block_counts<false>, dim3(grid_for(nblk + 1)), dim3(256), 0, ctx->stream, block_hist, nblk, shift, cnt,
                           (unsigned long long *) nullptr);
        ME_TRY(exclusive_scan_u32(ctx, cnt, block_off, nblk + 1));
    }
    const long long n_cells = c.level_unique[shift];
    ME_CHECK(ctx, t.cell_code.ensure((size_t) n_cells * 8));
    ME_CHECK(ctx, t.cell_start.ensure((size_t) (n_cells + 1) * 4));
    unsigned long long hsize = 64;
    while (hsize < 2ULL * (unsigned long long) n_cells) hsize <<= 1;
    ME_CHECK(ctx, t.hkeys.ensure((size_t) hsize * 8));
    ME_CHECK(ctx, t.hvals.ensure((size_t) hsize * 4));
    ME_CHECK(ctx, hipMemsetAsync(t.hkeys.p, 0xFF, (size_t) hsize * 8, ctx->stream));
    hipLaunchKernelGGL(k_cell_fill, dim3((unsigned int) nblk), dim3(256), 0, ctx->stream, c.codes.as<unsigned long long>(), n, 3 * shift,
                       block_off, n_cells, t.cell_code.as<unsigned long long>(), t.cell_start.as<unsigned int>(), t.hkeys.as<unsigned long long>(),
                       t.hvals.as<unsigned int>(), (unsigned int) (hsize - 1));
    t.shift = shift;
    g.cell_code = t.cell_code.as<unsigned long long>();
    g.cell_start = t.cell_start.as<unsigned int>();
    g.hkeys = t.hkeys.as<unsigned long long>();
    g.hvals = t.hvals.as<unsigned int>();
    g.hmask = (unsigned int) (hsize - 1);
    g.n_cells = n_cells;
    g.shift = shift;
    return ME_OK;
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
int cloud_upload(me_ctx *ctx, int slot, const double *src, bool src_on_device, long long n, const double *T,
                 double cell_size, bool prefiltered) {
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "slot must be ME_SLOT_EST (0) or ME_SLOT_GT (1)");
    if (prefiltered && n == 0) {  // this rank's slab (+ halo) holds nothing of this cloud: every pass returns empty partials
        ME_CHECK(ctx, hipSetDevice(ctx->device));
        Cloud &e = ctx->cloud[slot];
        e.n = e.n_total = 0;
        e.slab = ctx->slab;
        e.n_unres = 0;
        e.slab_identity = true;
        e.uploaded = true;
        e.index_valid = false;
        e.nn_ref_slot = -1;
        e.vox_valid = e.vox_merged = false;
        e.have_normals = e.have_cov = false;
        ctx->cloud[1 - slot].nn_ref_slot = -1;
        return ME_OK;
    }
    if (!src) return ctx->fail(ME_ERR_ARG, "xyz is NULL");
    if (n <= 0) return ctx->fail(ME_ERR_ARG, "point cloud is empty");  // map_eval.cpp:32-35 returns -1
    if (n >= (1LL << 31)) return ctx->fail(ME_ERR_ARG, "point count must be < 2^31 (the reference indexes with int)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    Cloud &c = ctx->cloud[slot];
    c.uploaded = false;
    c.index_valid = false;
    c.nn_ref_slot = -1;
    c.n_vox = 0;
    c.vox_size = 0;
    c.vox_valid = false;
    c.vox_merged = false;
    // any result that used this cloud as the reference is stale now
    ctx->cloud[1 - slot].nn_ref_slot = -1;
    c.n = n;
    c.n_total = n;
    c.have_normals = c.have_cov = false;
    c.slab = ctx->slab;
    c.n_unres = 0;
    c.slab_identity = true;  // (the filtered slab upload below clears it; a stale `false` would make me_slab_points read an old slab_orig)
    bool bbox_ready = false;
    // prefiltered: slab mode, but the caller guarantees that every point lies inside [reg_lo, reg_hi) (the halo exchange
    // delivered exactly those): the flag / scan / compact filter and its host round trip are skipped
    if (T) {  // the identity moves nothing (Open3D's Transform would multiply by 1 and add 0: bit-identical)
        bool identity = true;
        for (int i = 0; i < 16; ++i) identity = identity && T[i] == ((i % 5 == 0) ? 1.0 : 0.0);
        if (identity) T = nullptr;
    }
    if ((ctx->slab.axis < 0 || prefiltered) && src_on_device && !T && ctx->borrow_device_input) {
        // ME_FLAG_BORROW_DEVICE_INPUT: the cloud is read where it lies (the caller keeps the buffer valid and unchanged until this
        // slot's next upload): no 48-byte-per-point copy, the bounding box comes from k_bbox (cloud_finish)
        c.xyz.borrow(const_cast<double *>(src), (size_t) n * 3 * sizeof(double));
    } else if ((ctx->slab.axis < 0 || prefiltered) && src_on_device) {
        // one pass: copy + transform + bounding-box partials
        ME_CHECK(ctx, c.xyz.ensure((size_t) n * 3 * sizeof(double)));
        const unsigned int nb = (unsigned int) std::min<long long>(1024, (n + 255) / 256);
        ME_CHECK(ctx, ctx->red.ensure((size_t) nb * 6 * sizeof(double)));
        Mat4 m{};
        if (T) std::memcpy(m.m, T, sizeof(m.m));
        hipLaunchKernelGGL(k_ingest, dim3(nb), dim3(256), 0, ctx->stream, src, n, T ? 1 : 0, m, c.xyz.as_mut<double>(),
                           ctx->red.as<double>());
        bbox_ready = true;
    } else if (ctx->slab.axis < 0) {
        ME_CHECK(ctx, c.xyz.ensure((size_t) n * 3 * sizeof(double)));
        ME_TRY(copy_h2d(ctx, c.xyz.p, src, (size_t) n * 3 * sizeof(double)));
        if (T) {
            Mat4 m;
            std::memcpy(m.m, T, sizeof(m.m));
            hipLaunchKernelGGL(k_transform, dim3(grid_for(n)), dim3(256), 0, ctx->stream, c.xyz.as_mut<double>(), n, m);
        }
    } else {
        // slab mode: keep only [reg_lo, reg_hi) along the slab axis of the (transformed) cloud, stable order.  A device
        // buffer is filtered in place of a copy; a host buffer is staged first.
        DevBuf &stage = ctx->tmp[3], &flags = ctx->tmp[0], &pos = ctx->tmp[1];
        ME_CHECK(ctx, flags.ensure((size_t) n * 4));
        ME_CHECK(ctx, pos.ensure((size_t) n * 4));
        const double *in = src;
        if (!src_on_device) {
            ME_CHECK(ctx, stage.ensure((size_t) n * 3 * sizeof(double)));
            ME_TRY(copy_h2d(ctx, stage.p, src, (size_t) n * 3 * sizeof(double)));
            in = stage.as<double>();
        }
        Mat4 m{};
        if (T) std::memcpy(m.m, T, sizeof(m.m));
        TimerScope ts(ctx, "slab_filter");
        hipLaunchKernelGGL(k_slab_flags, dim3(grid_for(n)), dim3(256), 0, ctx->stream, in, n, T ? 1 : 0, m, c.slab,
                           flags.as<unsigned int>());
        ME_TRY(exclusive_scan_u32(ctx, flags.as<unsigned int>(), pos.as<unsigned int>(), n));
        unsigned int last_pos = 0, last_flag = 0;
        ME_TRY(copy_d2h(ctx, &last_pos, pos.as<unsigned int>() + (n - 1), 4));
        ME_TRY(copy_d2h(ctx, &last_flag, flags.as<unsigned int>() + (n - 1), 4));
        ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const long long kept = (long long) last_pos + last_flag;
        ME_CHECK(ctx, c.xyz.ensure((size_t) std::max<long long>(kept, 1) * 3 * sizeof(double)));
        ME_CHECK(ctx, c.slab_orig.ensure((size_t) std::max<long long>(kept, 1) * 4));
        c.slab_identity = false;
        hipLaunchKernelGGL(k_slab_compact, dim3(grid_for(n)), dim3(256), 0, ctx->stream, in, n, T ? 1 : 0, m,
                           flags.as<unsigned int>(), pos.as<unsigned int>(), c.xyz.as_mut<double>(), c.slab_orig.as<int>());
        c.n = kept;
        if (kept == 0) {  // this rank's slab (+halo) holds nothing of this cloud: every pass returns empty partials
            c.uploaded = true;
            c.index_valid = false;
            ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            return ME_OK;
        }
    }
    c.cell_size_req = cell_size;
    return cloud_finish(ctx, slot, bbox_ready);
}

// bbox of the points now in c.xyz, then the index (shared by upload, in-place down-sampling and in-place transform)
int cloud_finish(me_ctx *ctx, int slot, bool bbox_ready) {
    Cloud &c = ctx->cloud[slot];
    const long long n = c.n;
    c.uploaded = false;
    c.index_valid = false;
    c.nn_ref_slot = -1;
    c.vox_valid = false;
    c.vox_merged = false;
    c.n_vox = 0;
    ctx->cloud[1 - slot].nn_ref_slot = -1;
    // bbox
    const unsigned int nb = (unsigned int) std::min<long long>(1024, (n + 255) / 256);
    ME_CHECK(ctx, ctx->red.ensure((size_t) nb * 6 * sizeof(double)));
    if (!bbox_ready)  // (an upload from a device buffer has produced the partials in its single pass, k_ingest)
        hipLaunchKernelGGL(k_bbox, dim3(nb), dim3(256), 0, ctx->stream, c.xyz.as<double>(), n, ctx->red.as<double>());
    std::vector<double> part((size_t) nb * 6);
    {
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, part.data(), ctx->red.p, part.size() * sizeof(double)));
        ME_TRY(mg.sync());
    }
    for (int d = 0; d < 3; ++d) {
        double lo = INFINITY, hi = -INFINITY;
        for (unsigned int b = 0; b < nb; ++b) {
            lo = std::fmin(lo, part[6 * b + d]);
            hi = std::fmax(hi, part[6 * b + 3 + d]);
        }
        if (!std::isfinite(lo) || !std::isfinite(hi))
            return ctx->fail(ME_ERR_ARG, "cloud contains NaN/inf coordinates (the reference strips them at load, map_eval.cpp:6)");
        c.bbox_lo[d] = lo;
        c.bbox_hi[d] = hi;
    }
    // slab mode: whatever the upload path was (filter pass, or me_upload_slab_device which trusts the caller's exchange), every
    // point held must lie inside [reg_lo, reg_hi) along the slab axis — otherwise the halo the per-point passes rely on is not the
    // one me_set_slab declared (a caller whose cuts or halo differ from its exchange's would get silently incomplete neighbourhoods)
    if (c.slab.axis >= 0 && (c.bbox_lo[c.slab.axis] < c.slab.reg_lo || !(c.bbox_hi[c.slab.axis] < c.slab.reg_hi)))
        return ctx->fail(ME_ERR_ARG, "slab upload: points outside [lo - halo, hi + halo) of me_set_slab along the slab axis");
    c.uploaded = true;
    return cloud_build_index(ctx, slot, c.cell_size_req);
}

// *map_3d_ = map_3d_->Transform(T) (map_eval.cpp:1206) on the cloud already on the device
int cloud_transform(me_ctx *ctx, int slot, const double *T) {
    if (slot < 0 || slot > 1 || !T) return ctx->fail(ME_ERR_ARG, "me_transform_cloud: bad argument");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "cloud not uploaded");
    if (c.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_transform_cloud: not available in slab mode (pass T to the upload)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    if (c.n == 0) return ME_OK;
    {   // the identity moves nothing (Open3D's Transform would multiply by 1 and add 0: bit-identical): keep the index
        bool identity = true;
        for (int i = 0; i < 16; ++i) identity = identity && T[i] == ((i % 5 == 0) ? 1.0 : 0.0);
        if (identity) return ME_OK;
    }
    Mat4 m;
    std::memcpy(m.m, T, sizeof(m.m));
    ME_CHECK(ctx, c.xyz.make_owned(ctx->stream));  // (a borrowed input buffer is the caller's: transform a private copy)
    hipLaunchKernelGGL(k_transform, dim3(grid_for(c.n)), dim3(256), 0, ctx->stream, c.xyz.as_mut<double>(), c.n, m);
    ME_TRY(rotate_attributes(ctx, slot, T));  // normals / covariances follow the points (Open3D PointCloud::Transform)
    return cloud_finish(ctx, slot);
}

// Open3D PointCloud::Transform (map_eval.cpp:1206) on a raw device buffer, in place: what a rank applies to its part of the
// estimated map BEFORE the halo exchange, so that the slab a point lands in is decided on the exact transformed coordinate
int transform_points_device(me_ctx *ctx, double *xyz_device, long long n, const double *T) {
    if (n < 0 || (n > 0 && !xyz_device) || !T) return ctx->fail(ME_ERR_ARG, "me_transform_points_device: bad argument");
    if (n == 0) return ME_OK;
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    Mat4 m;
    std::memcpy(m.m, T, sizeof(m.m));
    hipLaunchKernelGGL(k_transform, dim3(grid_for(n)), dim3(256), 0, ctx->stream, xyz_device, n, m);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

// --- sparse octree above the 1-NN cells (general 1-NN path: k_nn1 / k_nn_far, me_nn.hip) ---
// small_top: the one-block kernel of the upper levels with four wavefronts instead of sixteen (a launch beside another lane's
// full-chip kernels, see below)
static int build_octree(me_ctx *ctx, Cloud &c, int nn_shift, bool small_top) {
    const long long n = c.n;
    OctView &v = c.oct;
    int L = 0;
    long long off = 0;
    for (int k = nn_shift; k <= kMortonBits; ++k) {
        if (L >= kMaxLevels) return ctx->fail(ME_ERR_ARG, "octree deeper than the level table (cloud extent / cell size too large)");
        v.count[L] = c.level_unique[k];
        v.off[L] = off;
        off += v.count[L] + 1;
        ++L;
        if (c.level_unique[k] == 1) break;
    }
    v.n_levels = L;
    ME_CHECK(ctx, c.oct_nodes.ensure((size_t) (off + kFan) * sizeof(ONode)));  // + slack for the 8-record burst
    ME_CHECK(ctx, hipMemsetAsync(c.oct_nodes.as<ONode>() + off, 0, kFan * sizeof(ONode), ctx->stream));
    ME_CHECK(ctx, c.oct_pbegin.ensure((size_t) (off + kFan + 1) * 4));
    ME_CHECK(ctx, hipMemsetAsync(c.oct_pbegin.as<unsigned int>() + off, 0, (kFan + 1) * 4, ctx->stream));
    v.nodes = c.oct_nodes.as<ONode>();
    v.pbegin = c.oct_pbegin.as<unsigned int>();
    ONode *nodes = c.oct_nodes.as<ONode>();
    unsigned int *pbeg = c.oct_pbegin.as<unsigned int>();
    hipLaunchKernelGGL(k_oct_leaves, dim3(grid_for((v.count[0] + 1) * 8)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(),
                       c.nn_grid.cell_start, v.count[0], n, nodes, pbeg);
    // prefix codes of the current level (level 0: the cell codes of the 1-NN grid), ping-pong
    DevBuf &ca = ctx->tmp[2], &cb = ctx->tmp[3], &pos = ctx->tmp[1], &begin = ctx->tmp[4];
    const unsigned long long *cur = c.nn_grid.cell_code;
    for (int l = 0; l + 1 < L; ++l) {
        const long long nc = v.count[l], np = v.count[l + 1];
        if (nc <= kOctTopMax) {  // everything from here up in ONE launch
            ME_CHECK(ctx, begin.ensure((size_t) (np + 1) * 4));
            ME_CHECK(ctx, ca.ensure((size_t) np * 8));
            ME_CHECK(ctx, cb.ensure((size_t) np * 8));
            // (`cur` may live in ca / cb: the kernel's first output buffer must be the other one)
            unsigned long long *first = (cur == ca.as<unsigned long long>()) ? cb.as<unsigned long long>() : ca.as<unsigned long long>();
            unsigned long long *second = (first == ca.as<unsigned long long>()) ? cb.as<unsigned long long>() : ca.as<unsigned long long>();
            // (a twin lane runs beside the main lane's full-chip kernels: a 1024-thread block needs 16 free wave slots on ONE CU
            // at the same moment and waited 10 ms for them under k_mme3 — profiles/r06_timeline.txt; four waves find room)
            if (small_top)
                hipLaunchKernelGGL((k_oct_top<256>), dim3(1), dim3(256), 0, ctx->stream, nodes, pbeg, v, l, cur, first, second,
                                   begin.as<unsigned int>());
            else
                hipLaunchKernelGGL((k_oct_top<1024>), dim3(1), dim3(1024), 0, ctx->stream, nodes, pbeg, v, l, cur, first, second,
                                   begin.as<unsigned int>());
            break;
        }
        ME_CHECK(ctx, pos.ensure((size_t) nc * 4));
        ME_CHECK(ctx, begin.ensure((size_t) (np + 1) * 4));
        DevBuf &nxt = (l % 2 == 0) ? ca : cb;
        ME_CHECK(ctx, nxt.ensure((size_t) np * 8));
        ME_TRY(cell_start_ranks(ctx, cur, nc, 3, pos.as<unsigned int>()));
        hipLaunchKernelGGL(k_cell_scatter, dim3(grid_for(nc)), dim3(256), 0, ctx->stream, cur, pos.as<unsigned int>(), nc, 3,
                           nxt.as<unsigned long long>(), begin.as<unsigned int>());
        hipLaunchKernelGGL(k_oct_up, dim3(grid_for(np + 1)), dim3(256), 0, ctx->stream, nodes + v.off[l], nc,
                           begin.as<unsigned int>(), np, nodes + v.off[l + 1], pbeg + v.off[l], pbeg + v.off[l + 1]);
        cur = nxt.as<unsigned long long>();
    }
    return ME_OK;
}

int cloud_build_index(me_ctx *ctx, int slot, double cell_size) {
    Cloud &c = ctx->cloud[slot];
    ME_TRACE_POINT(ctx, slot == 0 ? "cloud_build_index(est): enter" : "cloud_build_index(gt): enter");
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "cloud not uploaded");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    c.cell_size_req = cell_size;
    const long long n = c.n;
    double extent = 0;
    for (int d = 0; d < 3; ++d) extent = std::fmax(extent, c.bbox_hi[d] - c.bbox_lo[d]);
    if (!(cell_size > 0)) cell_size = (extent > 0 ? extent / 128.0 : 1.0);
    // a hair larger than the radius so that |p-q| < r can never straddle two cell boundaries through rounding
    const double cell_h = cell_size * (1.0 + 0x1p-20);
    // origin snapped to the global cell lattice: two clouds built with the same cell size get ALIGNED grids, so the
    // points of one cell of the query cloud fall into one cell of the reference cloud (the 1-NN grid pass groups
    // lanes by reference cell; with a common lattice a wave of curve-consecutive queries touches few of them)
    for (int d = 0; d < 3; ++d) c.origin[d] = std::floor(c.bbox_lo[d] / cell_h) * cell_h;
    extent = 0;
    for (int d = 0; d < 3; ++d) extent = std::fmax(extent, c.bbox_hi[d] - c.origin[d]);
    const double ncell = std::floor(extent / cell_h) + 1.0;
    int bits_cell = 1;
    while ((double) (1LL << bits_cell) < ncell && bits_cell <= kMortonBits) ++bits_cell;
    if (bits_cell > kMortonBits)
        return ctx->fail(ME_ERR_ARG, "cell size too small for the cloud extent (needs > 2^21 cells per axis)");
    c.shift = kMortonBits - bits_cell;
    c.cell_h = cell_h;
    c.fine_h = std::ldexp(cell_h, -c.shift);
    c.index_valid = false;
    c.mme_have = false;  // (the sorted order changes)
#ifdef ME_AB
    c.mme_feat_valid = false;
#endif
    c.nn_ref_slot = -1;
    ctx->cloud[1 - slot].nn_ref_slot = -1;

    // --- Morton codes + sort ---
    DevBuf &codes_in = ctx->tmp[0], &iota = ctx->tmp[1], &perm = ctx->tmp[2];
    ME_CHECK(ctx, codes_in.ensure((size_t) n * 8));
    ME_CHECK(ctx, iota.ensure((size_t) n * 4));
    ME_CHECK(ctx, perm.ensure((size_t) n * 4));
    ME_CHECK(ctx, c.codes.ensure((size_t) n * 8));
    ME_CHECK(ctx, c.sp.ensure((size_t) n * sizeof(SPoint)));
    // levels below the search cell that the order (and therefore the choice of the 1-NN grid) may use
    constexpr int sort_depth = ME_TUNE_SORT_DEPTH;
    // ME_FLAG_MORTON_ORDER: points sorted along the Z curve (the first version) — tests / A-B measurements
    const int hilbert = ctx->morton_order ? 0 : 1;
    // Keys-only sort when key and index fit one 64-bit word (round 4, k_morton): 3 (21 - min_level) key bits + ceil(log2 n) index
    // bits.  A 50 M-point cloud at a 0.1 m cell in a 200 m scene needs 33 + 3 depth + 26: depth 1 fits, depth 2 does not.  The depth
    // is lowered to make it fit — and if the 1-NN grid then lands on the finest sorted level (a dense cloud, which might have chosen
    // a finer one had it been sorted), the index is rebuilt with the pair sort at the full depth and the slot remembers it.
    int idx_bits = 1;
    while ((1LL << idx_bits) < n) ++idx_bits;
    constexpr int pack_allowed = ME_TUNE_SORT_PACK;
    int sort_min_level = std::max(0, c.shift - std::max(0, sort_depth));
    int pack_bits = 0;
    // (the hint of an earlier build of this slot holds for the SAME cloud shape only — point count, cell size, lattice depth, e.g. the
    // same cloud uploaded again — so that one dense cloud does not leave every later cloud of the slot on the slower pair sort)
    const bool pairs_hinted = c.sort_pairs_hint && c.hint_n == n && c.hint_shift == c.shift && c.hint_cell_h == cell_h;
    if (!pairs_hinted) c.sort_pairs_hint = false;
    if (pack_allowed && !pairs_hinted) {
        // (the depth is cut to 1 at most: at depth 0 the finest sorted level is the search cell itself, which holds its 6 points
        // in any cloud worth indexing — the rebuild below would be certain)
        const int lvl_max = std::max(sort_min_level, c.shift - std::min(std::max(0, sort_depth), 1));
        int lvl = sort_min_level;
        while (lvl < lvl_max && 3 * (kMortonBits - lvl) + idx_bits > 64) ++lvl;
        if (3 * (kMortonBits - lvl) + idx_bits <= 64) {
            sort_min_level = lvl;
            pack_bits = idx_bits;
        }
    }
    const bool depth_cut = pack_bits > 0 && sort_min_level > std::max(0, c.shift - std::max(0, sort_depth));
#ifdef ME_DBG_INDEX
    std::fprintf(stderr, "[index slot %d] n=%lld cell_h=%.17g origin=(%.17g %.17g %.17g) bbox_lo=(%.17g %.17g %.17g) shift=%d sort_min_level=%d pack_bits=%d hinted=%d\n", slot, n, cell_h,
                 c.origin[0], c.origin[1], c.origin[2], c.bbox_lo[0], c.bbox_lo[1], c.bbox_lo[2], c.shift, sort_min_level, pack_bits, (int) pairs_hinted);
#endif
    if (pack_bits > 0) ME_CHECK(ctx, perm.ensure((size_t) n * 8));  // (the sorted words)
    {
        TimerScope ts(ctx, "morton");
        hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(256), 0, ctx->stream, c.xyz.as<double>(), n, c.origin[0],
                           c.origin[1], c.origin[2], c.fine_h, sort_min_level, hilbert, pack_bits, codes_in.as<unsigned long long>(),
                           iota.as<unsigned int>());
    }
    // Nothing below consumes the order inside a cell 4x finer than the search cell: the 1-NN grid (and the octree leaves) is
    // chosen among the SORTED levels, and a quarter-cell that holds the 6 points it aims at means > 380 points in the search
    // cell (38 000 pts/m^2 of surface at r = 0.1 m); denser clouds simply get more points per 1-NN cell.  So the radix sort
    // skips the low bits: 5 passes instead of 8 on the bench scene (6 with the five levels sorted at first).
    // The sort is stable, so the order inside such a cell is the input order: still deterministic.
    if (pack_bits > 0)
        ME_TRY(sort_keys_u64(ctx, codes_in.as<unsigned long long>(), perm.as<unsigned long long>(), n, pack_bits,
                             pack_bits + 3 * (kMortonBits - sort_min_level)));
    else
        ME_TRY(sort_pairs_u64_u32(ctx, codes_in.as<unsigned long long>(), c.codes.as<unsigned long long>(),
                                  iota.as<unsigned int>(), perm.as<unsigned int>(), n, 3 * sort_min_level, 63));
    if (ctx->sort_hook) ctx->sort_hook(ctx->sort_hook_arg, ctx->stream);  // (me_run_suite_from: see SuiteLane::sort_queued)
    // voxel run records on the side, when the context carries a voxel size for them (me_run_suite_from) and their sort key fits
    VoxPack vp{};
    c.vox_rec_valid = false;
    const bool fuse_vox = ME_TUNE_VOX_ONEPASS && ctx->vox_hint > 0 && vox_make_pack(c, ctx->vox_hint, n, vp);
    const long long rec_cap = vox_record_capacity(n);
    if (fuse_vox) {
        ME_CHECK(ctx, c.vox_rec_key.ensure((size_t) rec_cap * 8));
        ME_CHECK(ctx, c.vox_rec_n.ensure((size_t) rec_cap * 4));
        ME_CHECK(ctx, c.vox_rec_s.ensure((size_t) rec_cap * 8 * kVoxRec));
        ME_CHECK(ctx, c.vox_rec_cnt.ensure(kVoxCounterBytes));
        ME_CHECK(ctx, hipMemsetAsync(c.vox_rec_key.p, 0xFF, (size_t) rec_cap * 8, ctx->stream));  // kVoxEmptySlot
        ME_CHECK(ctx, hipMemsetAsync(c.vox_rec_cnt.p, 0, kVoxCounterBytes, ctx->stream));
    }
    {
        TimerScope ts(ctx, "gather");
#define ME_LAUNCH_GATHER(VOX_, PER_)                                                                                                     \
    hipLaunchKernelGGL((k_gather<VOX_, PER_>), dim3(grid_for(n, 256 * PER_)), dim3(256), 0, ctx->stream, c.xyz.as<double>(),            \
                       pack_bits > 0 ? (const unsigned int *) nullptr : perm.as<unsigned int>(),                                       \
                       pack_bits > 0 ? perm.as<unsigned long long>() : (const unsigned long long *) nullptr,                           \
                       pack_bits > 0 ? ((1ULL << pack_bits) - 1ULL) : 0ULL, n, c.origin[0], c.origin[1], c.origin[2], c.fine_h,        \
                       c.sp.as<SPoint>(), c.codes.as<unsigned long long>(), vp, c.slab, c.vox_rec_key.as<unsigned long long>(),        \
                       c.vox_rec_n.as<int>(), c.vox_rec_s.as<double>(), c.vox_rec_cnt.as<unsigned int>(), (unsigned int) ((n + 63) / 64), \
                       (unsigned int) vox_region_size(n))
        if (fuse_vox) ME_LAUNCH_GATHER(true, kGatherPerVox);
        else ME_LAUNCH_GATHER(false, kGatherPer);
#undef ME_LAUNCH_GATHER
    }
    if (fuse_vox) {
        c.vox_rec_valid = true;
        c.vox_rec_cap = rec_cap;
        c.vox_rec_pack = vp;
    }
    // --- occupied cells per Morton level -> pick the 1-NN grid level; build the cell tables ---
    {
        TimerScope ts(ctx, "cells");
        // [hist 64 x u64 | block offsets | block counts | block histogram rows]: one pass over the sorted codes leaves a histogram row per
        // block; the rows are added up and the radius grid's cell counts (its level is known before the histogram is) scanned
        const int nblk = (int) ((n + kCellChunk - 1) / kCellChunk);
        ME_CHECK(ctx, ctx->red.ensure(64 * 8 + (size_t) (nblk + 2) * 4 * (2 + kHistLevels)));
        unsigned long long *d_hist = ctx->red.as<unsigned long long>();
        unsigned int *d_boff = reinterpret_cast<unsigned int *>(d_hist + 64), *d_bcnt = d_boff + (nblk + 2), *d_bhist = d_bcnt + (nblk + 2);
        ME_CHECK(ctx, hipMemsetAsync(d_hist, 0, 32 * 8, ctx->stream));
        hipLaunchKernelGGL(k_level_hist_rows, dim3((unsigned int) nblk), dim3(256), 0, ctx->stream, c.codes.as<unsigned long long>(), n, d_bhist);
        hipLaunchKernelGGL(k_block_counts<true>, dim3(grid_for(nblk + 1)), dim3(256), 0, ctx->stream, d_bhist, nblk, c.shift, d_bcnt, d_hist);
        ME_TRY(exclusive_scan_u32(ctx, d_bcnt, d_boff, nblk + 1));
        unsigned long long h_hist[32];
        {
            MailGuard mg(ctx);
            ME_TRY(mail_post(ctx, h_hist, d_hist, sizeof(h_hist)));
            ME_TRY(mg.sync());
        }
        long long acc = 1;
        for (int k = kMortonBits; k >= 0; --k) {  // level k: cell edge fine_h * 2^k
            if (k < kMortonBits) acc += (long long) h_hist[k];
            c.level_unique[k] = (k == kMortonBits) ? 1 : acc;
        }
        // 1-NN grid: the finest level whose occupied cells hold >= 6 points on average (27-cell stencil ~ a few
        // hundred candidates per query at most, yet a guaranteed radius of one cell edge resolves almost all queries)
        int nn_shift = kMortonBits - 1;
        constexpr double nn_occ = ME_TUNE_NN_OCC;
        for (int k = sort_min_level; k < kMortonBits; ++k)  // (levels below sort_min_level are not sorted: counts meaningless)
            if ((double) n / (double) c.level_unique[k] >= nn_occ) {
                nn_shift = k;
                break;
            }
        if (depth_cut && nn_shift == sort_min_level) {
            // the finest sorted level holds >= 6 points per cell: with the full depth a finer one might have been chosen.  Build
            // again with the pair sort (and start with it next time this slot is indexed).
            c.sort_pairs_hint = true;
            c.hint_n = n;
            c.hint_shift = c.shift;
            c.hint_cell_h = cell_h;
            ts.end();
            return cloud_build_index(ctx, slot, c.cell_size_req);
        }
        ME_TRY(build_grid_table_counted(ctx, c, c.shift, c.grid_tab, c.grid, nblk, d_bhist, d_boff, true));
        c.n_mid = 0;
        if (nn_shift == c.shift) {
            c.nn_grid = c.grid;
        } else {
            ME_TRY(build_grid_table_counted(ctx, c, nn_shift, c.nn_tab, c.nn_grid, nblk, d_bhist, d_boff, false));
            // tables of the levels in between, for the 1-NN cascade (me_nn.hip): each is one scan over the rows + one pass over the codes
            for (int k = nn_shift + 1; k < c.shift && c.n_mid < Cloud::kMaxMid; ++k) {
                ME_TRY(build_grid_table_counted(ctx, c, k, c.mid_tab[c.n_mid], c.mid_grid[c.n_mid], nblk, d_bhist, d_boff, false));
                ++c.n_mid;
            }
        }
        // --- sparse octree above the 1-NN cells ---
        // Nothing but the 1-NN fallback walks it (k_nn1 / k_nn_far): a caller that runs other passes first (me_run_suite_from: the MME
        // needs the cell tables only) asks for the octree later, cloud_finish_octree — 1.2 ms of launch-sized kernels that otherwise
        // stand between the cell tables and the first full-chip kernel of the step.
        c.oct_deferred = false;
        c.oct_nn_shift = nn_shift;
        ts.end();  // ("cells": the tables the radius search and the 1-NN grid pass use; the octree has its own timer since round 6)
        if (ctx->defer_octree) {
            c.oct_deferred = true;
        } else {
            TimerScope to(ctx, "octree");
            ME_TRY(build_octree(ctx, c, nn_shift, ctx->is_twin));
        }
    }
    ME_CHECK(ctx, hipGetLastError());
    // The index is COMPLETE on the device when this returns ("every call is synchronous on return", mapeval_hip.h).  Through round 4
    // the cell tables and the octree — everything queued after the level histogram's host read — were still in flight here: harmless
    // on one stream, a data race for two lanes (me_twin): the other lane's k_mme3 / k_nn_grid started on ITS stream while this one's
    // k_cell_fill (then k_hash_insert) was still filling the table they probe.  Python's hand-over latency hid it; the C++ lanes of me_run_suite_from
    // (microseconds) did not: the second evaluation of a process (fresh buffers holding stale bytes instead of the previous,
    // identical table) spun in hash_lookup for minutes at 50 M points (profiles/EXPERIMENTS.md "Round 5").
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.index_valid = true;
    ME_TRACE_POINT(ctx, "cloud_build_index: done (host side)");
    return ME_OK;
}

// the octree of a cloud indexed with ctx->defer_octree: built now, complete on the device on return (no-op otherwise)
int cloud_finish_octree(me_ctx *ctx, int slot) {
    Cloud &c = ctx->cloud[slot];
    if (!c.oct_deferred || !c.index_valid) return ME_OK;  // (no index: the next build decides again)
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    {
        TimerScope ts(ctx, "octree");
        ME_TRY(build_octree(ctx, c, c.oct_nn_shift, true));
    }
    ME_CHECK(ctx, hipGetLastError());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.oct_deferred = false;
    return ME_OK;
}

__global__ void k_slab_points(const double *__restrict__ xyz, long long n, SlabView s, const int *__restrict__ orig,
                              long long *__restrict__ orig_out, unsigned char *__restrict__ owned_out) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (orig_out) orig_out[i] = orig ? (long long) orig[i] : i;
    if (owned_out) owned_out[i] = slab_owned(s, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]) ? 1 : 0;
}

// me_slab_points: which uploaded point each per-point output entry belongs to, and whether this rank owns it
int slab_points(me_ctx *ctx, int slot, int64_t *orig_index, uint8_t *owned, long long capacity, long long *count) {
    if (slot < 0 || slot > 1 || !count) return ctx->fail(ME_ERR_ARG, "me_slab_points: bad argument");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "me_slab_points: cloud not uploaded");
    *count = c.n;
    if ((!orig_index && !owned) || c.n == 0) return ME_OK;
    if (capacity < c.n) return ctx->fail(ME_ERR_CAPACITY, "me_slab_points: capacity too small (count returned)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf &oi = ctx->tmp[0], &ow = ctx->tmp[1];
    ME_CHECK(ctx, oi.ensure((size_t) c.n * 8));
    ME_CHECK(ctx, ow.ensure((size_t) c.n));
    hipLaunchKernelGGL(k_slab_points, dim3(grid_for(c.n)), dim3(256), 0, ctx->stream, c.xyz.as<double>(), c.n, c.slab,
                       c.slab_identity ? (const int *) nullptr : c.slab_orig.as<int>(), orig_index ? oi.as<long long>() : nullptr,
                       owned ? ow.as<unsigned char>() : nullptr);
    if (orig_index) ME_TRY(copy_d2h(ctx, orig_index, oi.p, (size_t) c.n * 8));
    if (owned) ME_TRY(copy_d2h(ctx, owned, ow.p, (size_t) c.n));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

}  // namespace me
