// me_mme.hip — Mean Map Entropy: radius neighbourhood -> covariance -> 0.5*ln(2*pi*e*det)
// (ComputeMeanMapEntropyUsingNormalTBB map_eval.cpp:1608-1737; the OpenMP and serial variants :1538-1606,
//  :1438-1535 compute the same thing with k >= 10 / k >= 5).
//
// Mapping.  Points are Morton-sorted on a grid whose cell edge is (a hair above) the search radius, so the
// neighbours of every point of a cell lie in the 3x3x3 block around it, and each cell is one contiguous run of
// the sorted array.  A wavefront owns 64 consecutive sorted points.  Per round (usually one or two per wave) it
//   * groups the lanes whose cell is within Chebyshev distance 2 of the leader's (normally the whole wave) and resolves the
//     group's cell box grown by one with one hash probe per (lane, slot) into a wave-private LDS run table
//     (wave_group_table),
//   * streams every non-empty run ONCE with WAVE-UNIFORM addresses (one fetch feeds all 64 lanes; the compiler turns
//     it into scalar loads, the candidate sits in SGPRs), every group lane testing it against its own query in fp64.
// rocprofv3 SQ counters put this kernel at ~80 % of the fp64 VALU issue rate: it is bound by the candidate tests, not by
// memory.  It is tuned to 64 VGPRs (8 waves per SIMD: the scalar fetches are L2 round trips that only other waves can
// hide); the run table lives in LDS and the 64-bit point index is rebuilt where needed for that reason, and the four
// registers the allocator still spills are touched in the prologue / epilogue only.  A per-lane walk (fewer candidates
// per lane) was measured slower (vector-L1 tag-lookup bound), half-radius cells with a 5x5x5 stencil likewise (same
// union, 6x the probes), sub-wave groups as well (profiles/README.md).
// The reference materialises index/distance vectors per query and gathers a 3xk matrix; here nothing is
// materialised: k, sum(p-q) and sum((p-q)(p-q)^T) are accumulated in registers about the QUERY as origin
// (|p-q| < r, so the one-pass covariance is as well conditioned as the reference's two-pass one).
// The query itself (d2 == 0, the element the reference erases at :1672-1673) contributes zero to both sums, so
// dropping it is `k - 1`.
#include <cmath>
#include <cstdlib>

#include "me_internal.hpp"

#ifndef ME_MME_DEPTH
#define ME_MME_DEPTH 2
#endif

namespace me {

// sum over the two 32-lane halves of a wave (lanes l and l ^ 32 end with the total, bit-identical)
__device__ __forceinline__ double half2_sum_d(double v) {
    unsigned int lo = (unsigned int) __double2loint(v), hi = (unsigned int) __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int) d[0], (int) c[0]) + __hiloint2double((int) d[1], (int) c[1]);
}

#ifdef ME_AB  // round 1's kernel, for A/B measurements only (make EXTRA=-DME_AB; ME_MME_V=1 selects it)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_mme(const SPoint *__restrict__ sp, const unsigned long long *__restrict__ codes, long long i_begin, long long i_end,
      GridView g, SlabView slab, double r2, int min_k, double *__restrict__ ent_s, unsigned char *__restrict__ valid_s,
      double *__restrict__ part_sum, long long *__restrict__ part_cnt, unsigned int xcd_chunk) {
    // XCD-aware chunking (see k_nn1): gridDim.x is a multiple of 8
    const unsigned int vb = xcd_virtual_block(blockIdx.x, gridDim.x, xcd_chunk);
    // (the index is rebuilt from this 32-bit offset where it is needed: one live register instead of two)
    const unsigned int loc = vb * blockDim.x + threadIdx.x;
    bool active = i_begin + (long long) loc < i_end;
    const int shift3 = 3 * g.shift;
    const int cell_lim = 1 << (kMortonBits - g.shift);

    double qx = 0, qy = 0, qz = 0;
    unsigned long long mycell = ~0ULL;
    if (active) {
        const long long i = i_begin + (long long) loc;
        const SPoint q = sp[i];
        qx = q.x;
        qy = q.y;
        qz = q.z;
        mycell = codes[i] >> shift3;
        if (!slab_owned(slab, qx, qy, qz)) {  // slab mode: halo points are neighbours only, never queries
            ent_s[i] = 0.0;
            valid_s[i] = 0;
            active = false;
        }
    }
    int k = 0;
    double s1x = 0, s1y = 0, s1z = 0;
    double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    bool done = !active;

    // r2e = r2 for the lanes of the current group, -1 for the others: the group predicate rides on the radius
    // compare, so the loop needs no exec juggling of its own (the scalar side of this kernel is nearly as busy as the
    // vector side: one pointer walk, four wave-uniform fetches in flight)
    auto test = [&](const SPoint &p, double r2e) {
        const double dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
        const double d2 = (dx * dx + dy * dy) + dz * dz;  // bit-identical to the CPU path (no FMA)
        if (d2 < r2e) {                                   // strict, nanoflann RadiusResultSet [upstream]
            ++k;
            s1x += dx;
            s1y += dy;
            s1z += dz;
            sxx = fma(dx, dx, sxx);
            sxy = fma(dx, dy, sxy);
            sxz = fma(dx, dz, sxz);
            syy = fma(dy, dy, syy);
            syz = fma(dy, dz, syz);
            szz = fma(dz, dz, szz);
        }
    };
    auto stream_run = [&](int cs, int ce, double r2e) {
        const SPoint *p = sp + cs;
        int j = cs;
        for (; j + 4 <= ce; j += 4, p += 4) {
            const SPoint p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
            test(p0, r2e);
            test(p1, r2e);
            test(p2, r2e);
            test(p3, r2e);
        }
        for (; j < ce; ++j, ++p) {
            const SPoint p0 = p[0];
            test(p0, r2e);
        }
    };

    // wave-shared candidate streams (scalar loads): every group lane tests the union of the group's cells.  The run
    // table of a round lives in LDS (not in registers: the kernel is tuned to 64 VGPRs = 8 waves per SIMD)
    __shared__ int2 s_tab[4][kGroupTab + 1];
    int2 *tab = s_tab[threadIdx.x >> 6];
    while (__ballot(!done)) {
        GroupBox bx;
        int nk = 0;
        const int lane = threadIdx.x & 63;
        const int cx = (int) compact21(mycell), cy = (int) compact21(mycell >> 1), cz = (int) compact21(mycell >> 2);
        const bool in = wave_group_table<1>(!done, cx, cy, cz, g, cell_lim, lane, tab, bx, &nk);
        const double r2e = in ? r2 : -1.0;
        wave_for_each_run(tab, nk, lane, [&](int cs, int ce, int) { stream_run(cs, ce, r2e); });
        if (in) done = true;
        __builtin_amdgcn_wave_barrier();
    }

    double H = 0.0;
    bool ok = false;
    if (active) {
        const int kk = k - 1;  // drop the query itself (map_eval.cpp:1672-1673)
        if (kk >= min_k) {     // (:1675 k >= 10, :1458 k >= 5)
            const double inv_k = 1.0 / (double) kk, inv_km1 = 1.0 / (double) (kk - 1);
            const double cxx = (sxx - s1x * s1x * inv_k) * inv_km1;
            const double cxy = (sxy - s1x * s1y * inv_k) * inv_km1;
            const double cxz = (sxz - s1x * s1z * inv_k) * inv_km1;
            const double cyy = (syy - s1y * s1y * inv_k) * inv_km1;
            const double cyz = (syz - s1y * s1z * inv_k) * inv_km1;
            const double czz = (szz - s1z * s1z * inv_k) * inv_km1;
            // Eigen 3x3 determinant (cofactor expansion along row 0)
            const double det = cxx * (cyy * czz - cyz * cyz) - cxy * (cxy * czz - cyz * cxz) + cxz * (cxy * cyz - cyy * cxz);
            const double h = 0.5 * log(2.0 * M_PI * M_E * det);  // ComputeEntropy (:1656)
            if (!isnan(h) && !isinf(h)) {                         // (:1692)
                H = h;
                ok = true;
            }
        }
        const long long i = i_begin + (long long) loc;
        ent_s[i] = H;                       // 0.0 where invalid (:1614)
        valid_s[i] = ok ? 1 : 0;
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double bs = block_sum_256(H, smd);
    const long long bc = block_sum_256_ll(ok ? 1LL : 0LL, smi);
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = bs;
        part_cnt[blockIdx.x] = bc;
    }
}

#endif  // ME_AB

#ifdef ME_MME_STATS
// build with -DME_MME_STATS (profiles/README.md): per launch, [0] wave rounds, [1] candidates streamed, [2] lanes served,
// [3] accepted (query, candidate) pairs, [4..6] rounds / candidates / lanes served of the rounds after a wave's first —
// printed to stderr by mme_run
__device__ unsigned long long g_mme_stat[8];
#endif

// ------------------------------------------------------------------------------------------------------------
// k_mme3 — the wave-shared candidate streams of k_mme with (a) an FP32 pre-test, (b) the per-run adjacency cull and
// (c) candidates delivered through a wave-private LDS tile instead of scalar fetches.
//
// (a) fp64 VALU instructions issue at half the rate of 32-bit ones on gfx950 and the exact test costs 9 of them per
// candidate (3 subtractions, 5 for ((dx*dx + dy*dy) + dz*dz), the compare) — for a lane that accepts ~12 % of what its
// wave streams.  When a run is staged, the lane that loads candidate p also writes p' = p - o (o = corner of the round's
// cell box) as (p'x, p'y, p'z, |p'|^2) in FP32.  With a = -2 (q - o) and T = r^2 - |q - o|^2 per lane and round, a
// candidate costs
//        u = fma(p'x, ax, fma(p'y, ay, fma(p'z, az, |p'|^2)))          (u - T ~ d^2 - r^2)
// and two compares, u < T - E ("inside for sure") and u < T + E ("perhaps"), E = 2^-12 h^2 bounding everything FP32 does
// to u - T (below).  Only when some lane sits in the band in between is that lane's exact fp64 test evaluated (a few
// candidates in a thousand), so `k` and the accepted set are exactly those of the fp64 test, and the moments are
// accumulated from the fp64 coordinates exactly as in k_mme: the results are bit-identical to its.
// Error bound (h = cell edge; the box spans <= 7 cells per axis, so |p'|, |q - o| <= 7h per axis, |a| <= 14h):
//   rounding of p' and a moves d^2 by <= 3 * 2 * (8h) * (14h 2^-24) = 672 * 2^-24 h^2;   |p'|^2 <= 147 h^2: 147;
//   the three FMAs (partial sums <= 441 h^2): 1323;   s = |a|^2 <= 588 h^2, three roundings, a quarter of it: 441;
//   T's own rounding: 147;   r^2 in FP32: 1.        Sum 2731 * 2^-24 h^2 < 4096 * 2^-24 h^2 = E.
// (b) wave_group_table<1, true>: runs whose cell is adjacent to no lane of the group are not streamed at all.
// (c) The first pre-test version fetched (FP32 record, fp64 point) per candidate with scalar loads: 48 bytes per candidate
// through the scalar data cache instead of 32, for a third less VALU work per candidate — it ran SLOWER than k_mme
// (29.8 vs 27.2 ms per step, more so with fewer waves per SIMD: the scalar path, not the VALU, was the limit).  Here a
// run is copied with one coalesced vector load per TILE points; the FP32 record is read back with one broadcast
// ds_read_b128 per candidate and the fp64 point (three ds_read_b64) only when some lane may accept it.
// ------------------------------------------------------------------------------------------------------------
template <int TILE, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_mme3(const SPoint *__restrict__ sp, const unsigned long long *__restrict__ codes, long long i_begin, long long i_end,
       GridView g, FrameView fr, SlabView slab, double r2, int min_k, double *__restrict__ ent_s,
       unsigned char *__restrict__ valid_s, double *__restrict__ part_sum, long long *__restrict__ part_cnt,
       unsigned int xcd_chunk, int dbg, double cell_h, float thr_lo, float thr_hi) {
    // cell_h = edge of a radius-grid cell; thr_lo / thr_hi = r^2 -+ E in FP32 (E = 2^-12 cell_h^2): kernel arguments, i.e.
    // scalar registers for the whole kernel (computed in the kernel they lived in VGPRs and were spilled around the loop).
    // dbg: profiling switches (profiles/README.md) — 1: no candidate streaming at all, 2: pre-test only, nothing accepted.
    static_assert(TILE * 16 >= kGroupRows * 4, "the row masks of the cull alias the FP32 tile");
    const unsigned int vb = xcd_virtual_block(blockIdx.x, gridDim.x, xcd_chunk);
    const unsigned int loc = vb * blockDim.x + threadIdx.x;
    bool active = i_begin + (long long) loc < i_end;
    const int shift3 = 3 * g.shift;
    const int cell_lim = 1 << (kMortonBits - g.shift);

    double qx = 0, qy = 0, qz = 0;
    unsigned long long mycell = ~0ULL;
    if (active) {
        const long long i = i_begin + (long long) loc;
        const SPoint q = sp[i];
        qx = q.x;
        qy = q.y;
        qz = q.z;
        mycell = codes[i] >> shift3;
        if (!slab_owned(slab, qx, qy, qz)) {
            ent_s[i] = 0.0;
            valid_s[i] = 0;
            active = false;
        }
    }
    bool done = !active;

    __shared__ int2 s_tab[4][kGroupTab + 1];
    __shared__ float4 s_tf[4][TILE];     // FP32 records of the staged run (the cull's row masks while the table is built)
    __shared__ double2 s_txy[4][TILE];   // its fp64 coordinates: (x, y) as one 16-byte record, z apart — two LDS reads per
    __shared__ double s_tz[4][TILE];     // accepted candidate instead of three (a ds_read_b64 costs a SIMD 8.7 issue cycles)
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));  // scalar: the wave's LDS bases stay out of VGPRs
    int2 *tab = s_tab[wv];
    float4 *tf = s_tf[wv];
    double2 *txy = s_txy[wv];
    double *tdz = s_tz[wv];
    const int lane = threadIdx.x & 63;

    double det_keep = 0.0;  // determinant of the neighbourhood covariance (valid when have_det)
    bool have_det = false;  // the query has at least min_k neighbours
    // A lane accumulates in exactly ONE round (the one whose group it belongs to), so the moments live inside the round:
    // nothing of them is alive while the next round's table is built.
#ifdef ME_MME_STATS
    int st_round = 0;
#endif
    while (__ballot(!done)) {
        GroupBox bx;
        int nk = 0;
        const int cx = (int) compact21(mycell), cy = (int) compact21(mycell >> 1), cz = (int) compact21(mycell >> 2);
        const bool in = wave_group_table<1, true>(!done, cx, cy, cz, g, cell_lim, lane, tab, bx, &nk,
                                                  reinterpret_cast<unsigned int *>(tf));
        // wave-uniform, and kept in scalar registers (there is no scalar fp64 arithmetic: computed once per round on the
        // vector unit, then moved over)
        const double ox = uniform_f64(fr.ox + (double) bx.x0 * cell_h), oy = uniform_f64(fr.oy + (double) bx.y0 * cell_h),
                     oz = uniform_f64(fr.oz + (double) bx.z0 * cell_h);
        const float ax = (float) (-2.0 * (qx - ox)), ay = (float) (-2.0 * (qy - oy)), az = (float) (-2.0 * (qz - oz));
        const float s = fmaf(az, az, fmaf(ay, ay, ax * ax));
        // (the group predicate rides on the thresholds: lanes outside the group accept nothing)
        const float t_hi = (in && dbg != 2) ? fmaf(-0.25f, s, thr_hi) : -INFINITY;
        const float t_lo = (in && dbg != 2) ? fmaf(-0.25f, s, thr_lo) : -INFINITY;
        int k = 0;
        double s1x = 0, s1y = 0, s1z = 0;
        double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
#ifdef ME_MME_STATS
        int st_cand = 0;
#endif
        auto test = [&](const float4 &c, int j) {
            const float u = fmaf(c.x, ax, fmaf(c.y, ay, fmaf(c.z, az, c.w)));
            const bool hi = u < t_hi;
            const unsigned long long mh = __ballot(hi);
            if (mh) {  // some lane may hold this candidate inside its radius
                bool acc = u < t_lo;
                // (both compares land in scalar register pairs: the band test is scalar work, no VALU instruction)
                if (__builtin_expect(mh != __ballot(acc), 0)) {  // a lane in the band: its exact test decides
                    asm volatile("; band: exact test" ::: "memory");     // (keeps the compiler from evaluating it always)
                    const double2 exy = txy[j];
                    const double ex = exy.x - qx, ey = exy.y - qy, ez = tdz[j] - qz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    acc = acc || (hi && d2 < r2);                        // strict, nanoflann RadiusResultSet [upstream]
                }
                if (acc) {
                    const double2 pxy = txy[j];
                    const double dx = pxy.x - qx, dy = pxy.y - qy, dz = tdz[j] - qz;
                    ++k;
                    s1x += dx;
                    s1y += dy;
                    s1z += dz;
                    sxx = fma(dx, dx, sxx);
                    sxy = fma(dx, dy, sxy);
                    sxz = fma(dx, dz, sxz);
                    syy = fma(dy, dy, syy);
                    syz = fma(dy, dz, syz);
                    szz = fma(dz, dz, szz);
                }
            }
        };
        if (dbg != 1) wave_for_each_run(tab, nk, lane, [&](int cs, int ce, int) {
            for (int base = cs; base < ce; base += TILE) {
                const int n = min(TILE, ce - base);
#ifdef ME_MME_STATS
                st_cand += n;
#endif
                if (lane < n) {
                    const SPoint p = sp[base + lane];
                    const double px = p.x - ox, py = p.y - oy, pz = p.z - oz;
                    const float fx = (float) px, fy = (float) py, fz = (float) pz;
                    // |p'|^2 of the ROUNDED coordinates (the error bound is stated for them), rounded once
                    const double w = ((double) fx * (double) fx + (double) fy * (double) fy) + (double) fz * (double) fz;
                    tf[lane] = make_float4(fx, fy, fz, (float) w);
                    txy[lane] = make_double2(p.x, p.y);
                    tdz[lane] = p.z;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int j = 0;
                // (two records in flight, not four: eight registers fewer is what keeps this kernel at 64 VGPRs without
                // spilling inside the run loop — the spills of the four-deep version were 3.2x the kernel's useful HBM traffic)
#if ME_MME_DEPTH == 2
                for (; j + 2 <= n; j += 2) {
                    const float4 c0 = tf[j], c1 = tf[j + 1];
                    test(c0, j);
                    test(c1, j + 1);
                }
#endif
                for (; j < n; ++j) {
                    const float4 c0 = tf[j];
                    test(c0, j);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the tile is overwritten by the next chunk
            }
        });
#ifdef ME_MME_STATS
        {
            int ka = in ? k - 1 : 0;
            for (int o = 32; o > 0; o >>= 1) ka += __shfl_xor(ka, o, 64);
            const int served = __popcll(__ballot(in));
            if (lane == 0) {
                atomicAdd(&g_mme_stat[0], 1ULL);
                atomicAdd(&g_mme_stat[1], (unsigned long long) st_cand);
                atomicAdd(&g_mme_stat[2], (unsigned long long) served);
                atomicAdd(&g_mme_stat[3], (unsigned long long) ka);
                if (st_round > 0) {  // what the rounds after a wave's first one cost and serve
                    atomicAdd(&g_mme_stat[4], 1ULL);
                    atomicAdd(&g_mme_stat[5], (unsigned long long) st_cand);
                    atomicAdd(&g_mme_stat[6], (unsigned long long) served);
                }
            }
            ++st_round;
        }
#endif
        if (in) {
            done = true;
            const int kk = k - 1;  // drop the query itself (map_eval.cpp:1672-1673)
            if (kk >= min_k) {     // (:1675 k >= 10, :1458 k >= 5)
                const double inv_k = 1.0 / (double) kk, inv_km1 = 1.0 / (double) (kk - 1);
                const double cxx = (sxx - s1x * s1x * inv_k) * inv_km1;
                const double cxy = (sxy - s1x * s1y * inv_k) * inv_km1;
                const double cxz = (sxz - s1x * s1z * inv_k) * inv_km1;
                const double cyy = (syy - s1y * s1y * inv_k) * inv_km1;
                const double cyz = (syz - s1y * s1z * inv_k) * inv_km1;
                const double czz = (szz - s1z * s1z * inv_k) * inv_km1;
                // Eigen 3x3 determinant (cofactor expansion along row 0)
                det_keep = cxx * (cyy * czz - cyz * cyz) - cxy * (cxy * czz - cyz * cxz) + cxz * (cxy * cyz - cyy * cxz);
                have_det = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // (the logarithm stays outside the round loop: inside, the compiler hoists its polynomial constants into VGPRs that
    // live across the candidate loop and spills them)
    double H = 0.0;
    bool ok = false;
    if (active) {
        if (have_det) {
            const double h = 0.5 * log(2.0 * M_PI * M_E * det_keep);  // ComputeEntropy (:1656); NaN for det < 0
            if (!isnan(h) && !isinf(h)) {                              // (:1692)
                H = h;
                ok = true;
            }
        }
        unsigned int loc_e = loc;
        asm volatile("" : "+v"(loc_e));  // (recomputed, not carried: the 64-bit index was spilled across the whole kernel)
        const long long i = i_begin + (long long) loc_e;
        ent_s[i] = H;                       // 0.0 where invalid (:1614)
        valid_s[i] = ok ? 1 : 0;
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double bs = block_sum_256(H, smd);
    const long long bc = block_sum_256_ll(ok ? 1LL : 0LL, smi);
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = bs;
        part_cnt[blockIdx.x] = bc;
    }
}

#ifdef ME_AB
// ------------------------------------------------------------------------------------------------------------
// k_mme3h (round 4, MEASURED AND NOT ADOPTED: 25.62 ms per step against k_mme3's 25.67 on the 50 M + 50 M pair, results identical;
// behind -DME_AB, ME_MME_V=8) — k_mme3 with the wavefront as 32 QUERIES x 2 CANDIDATE SLOTS instead of 64 queries x 1.
// k_mme3's time is its fp64 accumulation: 13 instructions for every candidate SOME lane accepts (~80 % of the stream), executed by
// 64 lanes of which ~12 accept.  The candidates a wave has to stream are the union of its queries' neighbourhoods, and the union of
// 32 curve-consecutive queries is smaller than that of 64 (their patch is 0.11 m instead of 0.16 m across, next to a radius of
// 0.1 m).  So a wave serves its 64 points in two passes of 32: lanes l and l + 32 stand for the same query, the lower half tests the
// even candidates of a staged tile and the upper half the odd ones (a two-address ds_read costs what a broadcast does,
// profiles/r04_issue_rates.txt), every step covers two candidates, and the halves' moments are added once per round.  Same
// accepted set, same exact band; the moments are the same fp64 sums in a different order (valid flags and counts identical,
// entropies to ~1e-14 as between any two summation orders).  Why it does not pay: the candidate set is quantised to whole cells —
// the cell box of 32 queries grown by one is ~4 x 4 surface cells against ~4.5 x 4.5 for 64, 1.26x fewer candidates, not the 1.6x of the
// continuous estimate — and the run table (12 % of k_mme3) is built twice per wavefront.
// ------------------------------------------------------------------------------------------------------------
template <int TILE, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_mme3h(const SPoint *__restrict__ sp, const unsigned long long *__restrict__ codes, long long i_begin, long long i_end,
       GridView g, FrameView fr, SlabView slab, double r2, int min_k, double *__restrict__ ent_s,
       unsigned char *__restrict__ valid_s, double *__restrict__ part_sum, long long *__restrict__ part_cnt,
       unsigned int xcd_chunk, int dbg, double cell_h, float thr_lo, float thr_hi) {
    // cell_h = edge of a radius-grid cell; thr_lo / thr_hi = r^2 -+ E in FP32 (E = 2^-12 cell_h^2): kernel arguments, i.e.
    // scalar registers for the whole kernel (computed in the kernel they lived in VGPRs and were spilled around the loop).
    // dbg: profiling switches (profiles/README.md) — 1: no candidate streaming at all, 2: pre-test only, nothing accepted.
    static_assert(TILE * 16 >= kGroupRows * 4, "the row masks of the cull alias the FP32 tile");
    const unsigned int vb = xcd_virtual_block(blockIdx.x, gridDim.x, xcd_chunk);
    const unsigned int loc = vb * blockDim.x + threadIdx.x;
    const int shift3 = 3 * g.shift;
    const int cell_lim = 1 << (kMortonBits - g.shift);
    __shared__ int2 s_tab[4][kGroupTab + 1];
    __shared__ float4 s_tf[4][TILE];     // FP32 records of the staged run (the cull's row masks while the table is built)
    __shared__ double2 s_txy[4][TILE];   // its fp64 coordinates: (x, y) as one 16-byte record, z apart — two LDS reads per
    __shared__ double s_tz[4][TILE];     // accepted candidate instead of three (a ds_read_b64 costs a SIMD 8.7 issue cycles)
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));  // scalar: the wave's LDS bases stay out of VGPRs
    int2 *tab = s_tab[wv];
    float4 *tf = s_tf[wv];
    double2 *txy = s_txy[wv];
    double *tdz = s_tz[wv];
    const int lane = threadIdx.x & 63;

    double det_keep = 0.0;  // determinant of the neighbourhood covariance (valid when have_det)
    bool have_det = false;  // the query has at least min_k neighbours
    // TWO PASSES of 32 queries x 2 candidate slots (see the header of this kernel): in pass p the lanes l and l + 32 both stand for
    // the query of lane 32 p + (l & 31); the lower half tests the even candidates of a tile, the upper half the odd ones.
    // (the pass's query is loaded at the start of the pass, not carried: this kernel lives on 64 registers)
    const int half = lane >> 5;
  for (int pass = 0; pass < 2; ++pass) {
    const long long qi = i_begin + (long long) (loc - (unsigned int) lane + (unsigned int) ((lane & 31) + 32 * pass));
    double qx = 0, qy = 0, qz = 0;
    unsigned long long mycell = ~0ULL;
    bool done = true;
    if (qi < i_end) {
        const SPoint q = sp[qi];
        qx = q.x;
        qy = q.y;
        qz = q.z;
        mycell = codes[qi] >> shift3;
        done = !slab_owned(slab, qx, qy, qz);  // slab mode: halo points are neighbours only, never queries
    }
    double det_pass = 0.0;
    bool have_pass = false;
    // A lane accumulates in exactly ONE round (the one whose group it belongs to), so the moments live inside the round:
    // nothing of them is alive while the next round's table is built.
#ifdef ME_MME_STATS
    int st_round = 0;
#endif
    while (__ballot(!done)) {
        GroupBox bx;
        int nk = 0;
        const int cx = (int) compact21(mycell), cy = (int) compact21(mycell >> 1), cz = (int) compact21(mycell >> 2);
        const bool in = wave_group_table<1, true>(!done, cx, cy, cz, g, cell_lim, lane, tab, bx, &nk,
                                                  reinterpret_cast<unsigned int *>(tf));
        // wave-uniform, and kept in scalar registers (there is no scalar fp64 arithmetic: computed once per round on the
        // vector unit, then moved over)
        const double ox = uniform_f64(fr.ox + (double) bx.x0 * cell_h), oy = uniform_f64(fr.oy + (double) bx.y0 * cell_h),
                     oz = uniform_f64(fr.oz + (double) bx.z0 * cell_h);
        const float ax = (float) (-2.0 * (qx - ox)), ay = (float) (-2.0 * (qy - oy)), az = (float) (-2.0 * (qz - oz));
        const float s = fmaf(az, az, fmaf(ay, ay, ax * ax));
        // (the group predicate rides on the thresholds: lanes outside the group accept nothing)
        const float t_hi = (in && dbg != 2) ? fmaf(-0.25f, s, thr_hi) : -INFINITY;
        const float t_lo = (in && dbg != 2) ? fmaf(-0.25f, s, thr_lo) : -INFINITY;
        int k = 0;
        double s1x = 0, s1y = 0, s1z = 0;
        double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
#ifdef ME_MME_STATS
        int st_cand = 0;
#endif
        auto test = [&](const float4 &c, int j) {
            const float u = fmaf(c.x, ax, fmaf(c.y, ay, fmaf(c.z, az, c.w)));
            const bool hi = u < t_hi;
            const unsigned long long mh = __ballot(hi);
            if (mh) {  // some lane may hold this candidate inside its radius
                bool acc = u < t_lo;
                // (both compares land in scalar register pairs: the band test is scalar work, no VALU instruction)
                if (__builtin_expect(mh != __ballot(acc), 0)) {  // a lane in the band: its exact test decides
                    asm volatile("; band: exact test" ::: "memory");     // (keeps the compiler from evaluating it always)
                    const double2 exy = txy[j];
                    const double ex = exy.x - qx, ey = exy.y - qy, ez = tdz[j] - qz;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    acc = acc || (hi && d2 < r2);                        // strict, nanoflann RadiusResultSet [upstream]
                }
                if (acc) {
                    const double2 pxy = txy[j];
                    const double dx = pxy.x - qx, dy = pxy.y - qy, dz = tdz[j] - qz;
                    ++k;
                    s1x += dx;
                    s1y += dy;
                    s1z += dz;
                    sxx = fma(dx, dx, sxx);
                    sxy = fma(dx, dy, sxy);
                    sxz = fma(dx, dz, sxz);
                    syy = fma(dy, dy, syy);
                    syz = fma(dy, dz, syz);
                    szz = fma(dz, dz, szz);
                }
            }
        };
        if (dbg != 1) wave_for_each_run(tab, nk, lane, [&](int cs, int ce, int) {
            for (int base = cs; base < ce; base += TILE) {
                const int n = min(TILE, ce - base), n4 = (n + 3) & ~3;
#ifdef ME_MME_STATS
                st_cand += n;
#endif
                if (lane >= n && lane < n4) tf[lane] = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
                if (lane < n) {
                    const SPoint p = sp[base + lane];
                    const double px = p.x - ox, py = p.y - oy, pz = p.z - oz;
                    const float fx = (float) px, fy = (float) py, fz = (float) pz;
                    // |p'|^2 of the ROUNDED coordinates (the error bound is stated for them), rounded once
                    const double w = ((double) fx * (double) fx + (double) fy * (double) fy) + (double) fz * (double) fz;
                    tf[lane] = make_float4(fx, fy, fz, (float) w);
                    txy[lane] = make_double2(p.x, p.y);
                    tdz[lane] = p.z;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int j = 0;
                // (two records in flight, not four: eight registers fewer is what keeps this kernel at 64 VGPRs without
                // spilling inside the run loop — the spills of the four-deep version were 3.2x the kernel's useful HBM traffic)
                // the tile is padded to a multiple of four with records no query accepts: two candidates per step and half
                for (; j < n4; j += 4) {
                    const float4 c0 = tf[j + half], c1 = tf[j + 2 + half];
                    test(c0, j + half);
                    test(c1, j + 2 + half);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the tile is overwritten by the next chunk
            }
        });
#ifdef ME_MME_STATS
        {
            int ka = in ? k - 1 : 0;
            for (int o = 32; o > 0; o >>= 1) ka += __shfl_xor(ka, o, 64);
            const int served = __popcll(__ballot(in));
            if (lane == 0) {
                atomicAdd(&g_mme_stat[0], 1ULL);
                atomicAdd(&g_mme_stat[1], (unsigned long long) st_cand);
                atomicAdd(&g_mme_stat[2], (unsigned long long) served);
                atomicAdd(&g_mme_stat[3], (unsigned long long) ka);
                if (st_round > 0) {  // what the rounds after a wave's first one cost and serve
                    atomicAdd(&g_mme_stat[4], 1ULL);
                    atomicAdd(&g_mme_stat[5], (unsigned long long) st_cand);
                    atomicAdd(&g_mme_stat[6], (unsigned long long) served);
                }
            }
            ++st_round;
        }
#endif
        // the two halves hold the moments over the even / the odd candidates of the same query: add them (every lane takes part)
        k += __shfl_xor(k, 32, 64);
        s1x = half2_sum_d(s1x);
        s1y = half2_sum_d(s1y);
        s1z = half2_sum_d(s1z);
        sxx = half2_sum_d(sxx);
        sxy = half2_sum_d(sxy);
        sxz = half2_sum_d(sxz);
        syy = half2_sum_d(syy);
        syz = half2_sum_d(syz);
        szz = half2_sum_d(szz);
        if (in) {
            done = true;
            const int kk = k - 1;  // drop the query itself (map_eval.cpp:1672-1673)
            if (kk >= min_k) {     // (:1675 k >= 10, :1458 k >= 5)
                const double inv_k = 1.0 / (double) kk, inv_km1 = 1.0 / (double) (kk - 1);
                const double cxx = (sxx - s1x * s1x * inv_k) * inv_km1;
                const double cxy = (sxy - s1x * s1y * inv_k) * inv_km1;
                const double cxz = (sxz - s1x * s1z * inv_k) * inv_km1;
                const double cyy = (syy - s1y * s1y * inv_k) * inv_km1;
                const double cyz = (syz - s1y * s1z * inv_k) * inv_km1;
                const double czz = (szz - s1z * s1z * inv_k) * inv_km1;
                // Eigen 3x3 determinant (cofactor expansion along row 0)
                det_pass = cxx * (cyy * czz - cyz * cyz) - cxy * (cxy * czz - cyz * cxz) + cxz * (cxy * cyz - cyy * cxz);
                have_pass = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (half == pass) {  // this pass served the lane's own query
        det_keep = det_pass;
        have_det = have_pass;
    }
  }
    const bool active = i_begin + (long long) loc < i_end;  // (not owned / too few neighbours: H = 0, valid = 0)
    // (the logarithm stays outside the round loop: inside, the compiler hoists its polynomial constants into VGPRs that
    // live across the candidate loop and spills them)
    double H = 0.0;
    bool ok = false;
    if (active) {
        if (have_det) {
            const double h = 0.5 * log(2.0 * M_PI * M_E * det_keep);  // ComputeEntropy (:1656); NaN for det < 0
            if (!isnan(h) && !isinf(h)) {                              // (:1692)
                H = h;
                ok = true;
            }
        }
        unsigned int loc_e = loc;
        asm volatile("" : "+v"(loc_e));  // (recomputed, not carried: the 64-bit index was spilled across the whole kernel)
        const long long i = i_begin + (long long) loc_e;
        ent_s[i] = H;                       // 0.0 where invalid (:1614)
        valid_s[i] = ok ? 1 : 0;
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double bs = block_sum_256(H, smd);
    const long long bc = block_sum_256_ll(ok ? 1LL : 0LL, smi);
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = bs;
        part_cnt[blockIdx.x] = bc;
    }
}


#endif  // ME_AB

#ifdef ME_AB
// ------------------------------------------------------------------------------------------------------------
// k_mme6 (round 3, MEASURED AND NOT ADOPTED: 33.8 ms per step against k_mme3's 26.1 on the 50 M + 50 M pair, results
// identical; profiles/README.md "round 3" has the counters) — the radius test on the MATRIX pipe, 16 queries x 4
// candidate slices per wavefront.
//
// u = |p'|^2 - 2 p'.q' over (candidates x queries) is a K = 4 contraction, (p'x, p'y, p'z, |p'|^2) . (ax, ay, az, 1):
// one v_mfma_f32_16x16x4_f32 (32 cycles on the matrix pipe, exact FP32 = an fmaf chain) evaluates it for 16 candidates x
// 16 queries, where the VALU version spent 4.5 instructions per candidate and 64 queries.  The output layout gives lane
// (s = lane / 16, j = lane % 16) four candidates of query j; the candidates of a 16-block are dealt so that result
// register r of slice s is candidate 4 r + s: an accumulation step (r) covers FOUR stream-consecutive candidates for the
// 16 (curve-consecutive, i.e. spatially clustered) queries, and is skipped when none of the 64 pairs is accepted.
// A wavefront owns 64 curve-consecutive points as before and serves them in four PASSES of 16 (every pass: all 64 lanes =
// 16 queries x 4 slices, its own cell box and run table — the box of 16 queries holds about half the candidates of the box
// of 64 —, candidate tiles packed ACROSS runs so that the MFMA blocks are full).  A slice accumulates k, sum(p - q),
// sum((p - q)(p - q)^T) over its quarter of the candidates; two half-exchange permutes per value add the four slices.
// The accepted set is exactly the fp64 test's (same exact band as k_mme3); the moments are the same fp64 sums in a different
// order (valid flags and counts bit-identical, entropies to ~1e-14).
// Error bound of the pre-test (h = cell edge; groups of Chebyshev radius 1: the box spans <= 5 cells, |p'|, |q'| <= 5h per
// axis, |a| <= 10h), in units of 2^-24 h^2: rounding of p' and a: 600;  |p'|^2 <= 75 h^2: 75;  the MFMA's chain (four
// roundings, partial sums <= 225 h^2): 900;  |a|^2 / 4 (<= 75 h^2, four roundings): 300;  T's rounding: 76;  r^2: 1.
// Sum 1952 < 4096 = E.
// ------------------------------------------------------------------------------------------------------------
typedef float mme_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMme6Tab = 125;  // (2 R + 1 + 2 H)^3, R = H = 1

template <int TILE, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_mme6(const SPoint *__restrict__ sp, const unsigned long long *__restrict__ codes, long long i_begin, long long i_end,
       GridView g, FrameView fr, SlabView slab, double r2, int min_k, double *__restrict__ ent_s,
       unsigned char *__restrict__ valid_s, double *__restrict__ part_sum, long long *__restrict__ part_cnt,
       unsigned int xcd_chunk, double cell_h, float thr_lo, float thr_hi) {
    static_assert(TILE == 32 || TILE == 64, "tile = 2 or 4 MFMA blocks");
    static_assert(TILE * 16 >= kGroupRows * 4, "the row masks of the cull alias the FP32 tile");
    constexpr int NBLK = TILE / 16;
    const unsigned int vb = xcd_virtual_block(blockIdx.x, gridDim.x, xcd_chunk);
    const int lane = threadIdx.x & 63;
    const int sl = lane >> 4, qj = lane & 15;  // slice, query of the pass
    const unsigned int wbase = vb * blockDim.x + (threadIdx.x & ~63u);  // the wave's first point (relative to i_begin)
    const int shift3 = 3 * g.shift;
    const int cell_lim = 1 << (kMortonBits - g.shift);

    __shared__ int2 s_tab[4][kMme6Tab + 3];
    __shared__ float4 s_tf[4][TILE];     // FP32 records of the staged tile (the cull's row masks while the table is built)
    __shared__ double s_td[4][3][TILE];  // its fp64 coordinates, one array per axis
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    int2 *tab = s_tab[wv];
    float4 *tf = s_tf[wv];
    double *tdx = s_td[wv][0], *tdy = s_td[wv][1], *tdz = s_td[wv][2];
    // A operand of block b: feature (lane / 16) of the candidate that MFMA row (lane % 16) stands for: row 4 s' + r' is
    // candidate 4 r' + s' of the block (so that result register r' of slice s' is that candidate)
    const float *tfa = reinterpret_cast<const float *>(tf) + (4 * (qj & 3) + (qj >> 2)) * 4 + sl;

    double det_mine = 0.0;   // of point (wave's first + lane): set in the pass that serves it (pass == slice)
    bool have_mine = false;
#ifdef ME_MME_STATS
    unsigned int st_rounds = 0, st_cand = 0, st_blocks = 0, st_slots = 0, st_pairs = 0, st_served = 0;
#endif
    for (int gp = 0; gp < 4; ++gp) {
        const long long i = i_begin + (long long) (wbase + 16u * gp + qj);
        bool act = i < i_end;
        double qx = 0, qy = 0, qz = 0;
        unsigned long long mycell = ~0ULL;
        if (act) {
            const SPoint q = sp[i];
            qx = q.x;
            qy = q.y;
            qz = q.z;
            mycell = codes[i] >> shift3;
            if (!slab_owned(slab, qx, qy, qz)) act = false;  // slab mode: halo points are neighbours only, never queries
        }
        bool done = !act;
        double det = 0.0;
        bool have = false;
        const int cx = (int) compact21(mycell), cy = (int) compact21(mycell >> 1), cz = (int) compact21(mycell >> 2);
        while (__ballot(!done)) {
            GroupBox bx;
            int nk = 0;
            const bool in = wave_group_table<1, true, 1>(!done, cx, cy, cz, g, cell_lim, lane, tab, bx, &nk,
                                                         reinterpret_cast<unsigned int *>(tf));
            const double ox = uniform_f64(fr.ox + (double) bx.x0 * cell_h), oy = uniform_f64(fr.oy + (double) bx.y0 * cell_h),
                         oz = uniform_f64(fr.oz + (double) bx.z0 * cell_h);
            const float ax = (float) (-2.0 * (qx - ox)), ay = (float) (-2.0 * (qy - oy)), az = (float) (-2.0 * (qz - oz));
            const float s2 = fmaf(az, az, fmaf(ay, ay, ax * ax));
            // (the group predicate rides on the thresholds: lanes outside the group accept nothing)
            const float t_hi = in ? fmaf(-0.25f, s2, thr_hi) : -INFINITY;
            const float t_lo = in ? fmaf(-0.25f, s2, thr_lo) : -INFINITY;
            const float bq = sl == 0 ? ax : (sl == 1 ? ay : (sl == 2 ? az : 1.0f));  // B operand: row (lane / 16) of (a, 1)
            int k = 0;
            double s1x = 0, s1y = 0, s1z = 0;
            double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;

            // one accumulation step: result u of this lane's candidate `c` (tile slot) against its query
            auto step = [&](float u, int c) {
                const bool hi = u < t_hi;
                const unsigned long long mh = __ballot(hi);
                if (mh) {  // some (query, candidate) pair of the step may be inside the radius
                    bool acc = u < t_lo;
                    if (__builtin_expect(mh != __ballot(acc), 0)) {  // a lane in the band: its exact test decides
                        asm volatile("; band: exact test" ::: "memory");
                        const double ex = tdx[c] - qx, ey = tdy[c] - qy, ez = tdz[c] - qz;
                        const double d2 = (ex * ex + ey * ey) + ez * ez;
                        acc = acc || (hi && d2 < r2);  // strict, nanoflann RadiusResultSet [upstream]
                    }
#ifdef ME_MME_STATS
                    ++st_slots;
                    st_pairs += (unsigned int) __popcll(__ballot(acc));
#endif
                    if (acc) {
                        const double dx = tdx[c] - qx, dy = tdy[c] - qy, dz = tdz[c] - qz;
                        ++k;
                        s1x += dx;
                        s1y += dy;
                        s1z += dz;
                        sxx = fma(dx, dx, sxx);
                        sxy = fma(dx, dy, sxy);
                        sxz = fma(dx, dz, sxz);
                        syy = fma(dy, dy, syy);
                        syz = fma(dy, dz, syz);
                        szz = fma(dz, dz, szz);
                    }
                }
            };
            // the first `nb` 16-blocks of the staged tile: all MFMAs first (independent), then their results
            auto process = [&](int nb) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                mme_f32x4 d[NBLK];
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
                    if (b < nb) {
                        const mme_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        d[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(tfa[b * 64], bq, zero, 0, 0, 0);
                    }
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
                    if (b < nb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) step(d[b][r], 16 * b + 4 * r + sl);
                    }
#ifdef ME_MME_STATS
                st_blocks += (unsigned int) nb;
#endif
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the tile is overwritten by the next one
            };
            int filled = 0;  // wave-uniform: slots of the tile already staged (tiles are packed across runs)
            wave_for_each_run(tab, nk, lane, [&](int cs, int ce, int) {
                int pos = cs;
                while (pos < ce) {
                    const int m = min(TILE - filled, ce - pos);
                    if (lane >= filled && lane < filled + m) {
                        const SPoint p = sp[pos + (lane - filled)];
                        const double px = p.x - ox, py = p.y - oy, pz = p.z - oz;
                        const float fx = (float) px, fy = (float) py, fz = (float) pz;
                        // |p'|^2 of the ROUNDED coordinates (the error bound is stated for them), rounded once
                        const double w = ((double) fx * (double) fx + (double) fy * (double) fy) + (double) fz * (double) fz;
                        tf[lane] = make_float4(fx, fy, fz, (float) w);
                        tdx[lane] = p.x;
                        tdy[lane] = p.y;
                        tdz[lane] = p.z;
                    }
                    filled += m;
                    pos += m;
#ifdef ME_MME_STATS
                    st_cand += (unsigned int) m;
#endif
                    if (filled == TILE) {
                        process(NBLK);
                        filled = 0;
                    }
                }
            });
            if (filled) {  // the last, partial tile: pad its last block with records no query accepts (u = +inf)
                const int up = (filled + 15) & ~15;
                if (lane >= filled && lane < up) tf[lane] = make_float4(0.f, 0.f, 0.f, INFINITY);
                process(up >> 4);
            }
            // add the four slices (every lane takes part: lanes outside the group hold zeros)
            k = rows4_sum_i(k);
            s1x = rows4_sum_d(s1x);
            s1y = rows4_sum_d(s1y);
            s1z = rows4_sum_d(s1z);
            sxx = rows4_sum_d(sxx);
            sxy = rows4_sum_d(sxy);
            sxz = rows4_sum_d(sxz);
            syy = rows4_sum_d(syy);
            syz = rows4_sum_d(syz);
            szz = rows4_sum_d(szz);
#ifdef ME_MME_STATS
            ++st_rounds;
            st_served += (unsigned int) (__popcll(__ballot(in)) >> 2);
#endif
            if (in) {
                done = true;
                const int kk = k - 1;  // drop the query itself (map_eval.cpp:1672-1673)
                if (kk >= min_k) {     // (:1675 k >= 10, :1458 k >= 5)
                    const double inv_k = 1.0 / (double) kk, inv_km1 = 1.0 / (double) (kk - 1);
                    const double cxx = (sxx - s1x * s1x * inv_k) * inv_km1;
                    const double cxy = (sxy - s1x * s1y * inv_k) * inv_km1;
                    const double cxz = (sxz - s1x * s1z * inv_k) * inv_km1;
                    const double cyy = (syy - s1y * s1y * inv_k) * inv_km1;
                    const double cyz = (syz - s1y * s1z * inv_k) * inv_km1;
                    const double czz = (szz - s1z * s1z * inv_k) * inv_km1;
                    // Eigen 3x3 determinant (cofactor expansion along row 0)
                    det = cxx * (cyy * czz - cyz * cyz) - cxy * (cxy * czz - cyz * cxz) + cxz * (cxy * cyz - cyy * cxz);
                    have = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (sl == gp) {  // lane (s, j) of pass s is point 16 s + j = this lane's own point
            det_mine = det;
            have_mine = have;
        }
    }
#ifdef ME_MME_STATS
    if (lane == 0) {
        atomicAdd(&g_mme_stat[0], (unsigned long long) st_rounds);
        atomicAdd(&g_mme_stat[1], (unsigned long long) st_cand);
        atomicAdd(&g_mme_stat[2], (unsigned long long) st_served);
        atomicAdd(&g_mme_stat[3], (unsigned long long) st_pairs);
        atomicAdd(&g_mme_stat[4], (unsigned long long) st_blocks);
        atomicAdd(&g_mme_stat[5], (unsigned long long) st_slots);
    }
#endif
    // (the logarithm stays outside the loops: inside, the compiler hoists its polynomial constants into VGPRs that live
    // across the candidate loop)
    double H = 0.0;
    bool ok = false;
    const long long i = i_begin + (long long) (wbase + (unsigned int) lane);
    if (i < i_end) {
        if (have_mine) {
            const double h = 0.5 * log(2.0 * M_PI * M_E * det_mine);  // ComputeEntropy (:1656); NaN for det < 0
            if (!isnan(h) && !isinf(h)) {                              // (:1692)
                H = h;
                ok = true;
            }
        }
        ent_s[i] = H;  // 0.0 where invalid (:1614) and for the halo points of a slab
        valid_s[i] = ok ? 1 : 0;
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double bs = block_sum_256(H, smd);
    const long long bc = block_sum_256_ll(ok ? 1LL : 0LL, smi);
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = bs;
        part_cnt[blockIdx.x] = bc;
    }
}
#endif  // ME_AB

__global__ void k_mme_unpermute(const SPoint *__restrict__ sp, long long i_begin, long long i_end,
                                const double *__restrict__ ent_s, const unsigned char *__restrict__ valid_s,
                                double *__restrict__ ent_o, unsigned char *__restrict__ valid_o) {
    const long long i = i_begin + (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i_end) return;
    const long long o = sp[i].idx;
    if (ent_o) ent_o[o] = ent_s[i];
    if (valid_o) valid_o[o] = valid_s[i];
}

// deterministic final reduction over block partials: block b sums the contiguous chunk [b*chunk, (b+1)*chunk) with a
// fixed tree; called twice (nb -> 256 -> 1) so that hundreds of thousands of partials do not serialise on one block
__global__ void __launch_bounds__(256)
k_mme_final(const double *__restrict__ ps, const long long *__restrict__ pc, long long nb, long long chunk,
            double *__restrict__ out_s, long long *__restrict__ out_c) {
    double s = 0;
    long long c = 0;
    const long long b0 = (long long) blockIdx.x * chunk, b1 = b0 + chunk < nb ? b0 + chunk : nb;
    for (long long b = b0 + threadIdx.x; b < b1; b += 256) {
        s += ps[b];
        c += pc[b];
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double rs = block_sum_256(s, smd);
    const long long rc = block_sum_256_ll(c, smi);
    if (threadIdx.x == 0) {
        out_s[blockIdx.x] = rs;
        out_c[blockIdx.x] = rc;
    }
}

int mme_run(me_ctx *ctx, int slot, double radius, int min_k, double *entropies, uint8_t *valid, double *sum_H,
            long long *n_valid) {
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    if (!(radius > 0)) return ctx->fail(ME_ERR_ARG, "me_mme: radius must be > 0");
    if (min_k < 2) return ctx->fail(ME_ERR_ARG, "me_mme: min_k must be >= 2 (covariance divides by k-1)");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "me_mme: cloud not uploaded");
    if (c.slab.axis >= 0 && c.slab.reg_lo > -INFINITY && c.slab.lo - c.slab.reg_lo < radius)
        return ctx->fail(ME_ERR_ARG, "me_mme: slab halo is smaller than the radius");
    if (c.n == 0) {  // empty slab
        if (sum_H) *sum_H = 0.0;
        if (n_valid) *n_valid = 0;
        return ME_OK;
    }
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    // the 27-cell stencil is exact only when cell edge >= radius; rebuild when it is not, or when the cells are
    // needlessly coarse (more candidates per query than necessary)
    const double want_h = radius * (1.0 + 0x1p-20);
    if (!c.index_valid || c.cell_h < want_h || c.cell_h > 1.5 * want_h) ME_TRY(cloud_build_index(ctx, slot, radius));
    const long long n = c.n;
    long long b, e;
    ctx->shard_range(n, b, e);
#ifdef ME_AB  // A/B build: ME_MME_V=7 runs round 4's matrix-pipe kernel (me_mme7.hip) where the cloud fits its fixed-point frame
    bool use7 = false;
    static const int force_v = std::getenv("ME_MME_V") ? std::atoi(std::getenv("ME_MME_V")) : 3;
    if (force_v == 7) ME_TRY(mme7_prepare(ctx, c, radius, &use7));
#endif
    DevBuf &ent_s = c.mme_ent, &val_s = c.mme_val;  // kept for me_render_entropy
    c.mme_have = false;
    ME_CHECK(ctx, ent_s.ensure((size_t) n * 8));
    ME_CHECK(ctx, val_s.ensure((size_t) n));
    const unsigned int nb = (unsigned int) std::max<long long>(8, ((e - b + 255) / 256 + 7) / 8 * 8);
    constexpr int kStage = 256;
    ME_CHECK(ctx, ctx->red.ensure((size_t) (nb + kStage + 1) * 16 + 64));
    double *ps = ctx->red.as<double>();
    long long *pc = reinterpret_cast<long long *>(ps + nb);
    double *ps2 = reinterpret_cast<double *>(pc + nb);
    long long *pc2 = reinterpret_cast<long long *>(ps2 + kStage);
    double *outs = reinterpret_cast<double *>(pc2 + kStage);
    long long *outc = reinterpret_cast<long long *>(outs + 1);
    const double r2 = radius * radius;  // Open3D SearchRadius -> nanoflann radiusSearch(q, r*r) [upstream]
    {
        const FrameView fr{c.origin[0], c.origin[1], c.origin[2], c.fine_h};
        const float band = (float) (0x1p-12 * c.cell_h * c.cell_h);  // E, see k_mme3
        const float thr_lo = (float) r2 - band, thr_hi = (float) r2 + band;
        TimerScope ts(ctx, "mme");
#define ME_LAUNCH_MME3(T, W, DBG)                                                                                             \
    hipLaunchKernelGGL((k_mme3<T, W>), dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(),                                \
                       c.codes.as<unsigned long long>(), b, e, c.grid, fr, c.slab, r2, min_k, ent_s.as<double>(),             \
                       val_s.as<unsigned char>(), ps, pc, xcd_chunk_setting(), DBG, c.cell_h, thr_lo, thr_hi)
#ifdef ME_AB
        // A/B build only (make EXTRA=-DME_AB): ME_MME_V=1 runs round 1's kernel (exact fp64 test per candidate, no cull),
        // 6 round 3's MFMA kernel; ME_MME_WAVES / ME_MME_TILE pick the compiled occupancy / tile, ME_MME_DBG the profiling cuts
        static const int variant = std::getenv("ME_MME_V") ? std::atoi(std::getenv("ME_MME_V")) : 3;
        static const int waves = std::getenv("ME_MME_WAVES") ? std::atoi(std::getenv("ME_MME_WAVES")) : 8;
        static const int tile = std::getenv("ME_MME_TILE") ? std::atoi(std::getenv("ME_MME_TILE")) : 32;
        static const int dbg = std::getenv("ME_MME_DBG") ? std::atoi(std::getenv("ME_MME_DBG")) : 0;
#define ME_LAUNCH_MME6(T, W)                                                                                                  \
    hipLaunchKernelGGL((k_mme6<T, W>), dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(),                                \
                       c.codes.as<unsigned long long>(), b, e, c.grid, fr, c.slab, r2, min_k, ent_s.as<double>(),             \
                       val_s.as<unsigned char>(), ps, pc, xcd_chunk_setting(), c.cell_h, thr_lo, thr_hi)
        if (variant == 8 && waves < 8) {
            hipLaunchKernelGGL((k_mme3h<32, 7>), dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), c.codes.as<unsigned long long>(), b, e,
                               c.grid, fr, c.slab, r2, min_k, ent_s.as<double>(), val_s.as<unsigned char>(), ps, pc, xcd_chunk_setting(), dbg,
                               c.cell_h, thr_lo, thr_hi);
        } else if (variant == 8) {
            hipLaunchKernelGGL((k_mme3h<32, 8>), dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), c.codes.as<unsigned long long>(), b, e,
                               c.grid, fr, c.slab, r2, min_k, ent_s.as<double>(), val_s.as<unsigned char>(), ps, pc, xcd_chunk_setting(), dbg,
                               c.cell_h, thr_lo, thr_hi);
        } else if (use7) {
            ME_TRY(mme7_launch(ctx, c, b, e, nb, radius, min_k, ent_s.as<double>(), val_s.as<unsigned char>(), ps, pc));
        } else if (variant == 1) {
            hipLaunchKernelGGL(k_mme, dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), c.codes.as<unsigned long long>(), b, e,
                               c.grid, c.slab, r2, min_k, ent_s.as<double>(), val_s.as<unsigned char>(), ps, pc, xcd_chunk_setting());
        } else if (variant == 6) {
            if (tile == 64) ME_LAUNCH_MME6(64, 6);
            else if (waves < 8) ME_LAUNCH_MME6(32, 6);
            else ME_LAUNCH_MME6(32, 8);
        } else if (tile == 32 && waves >= 8) ME_LAUNCH_MME3(32, 8, dbg);
        else if (tile == 32) ME_LAUNCH_MME3(32, 7, dbg);
        else if (waves >= 7) ME_LAUNCH_MME3(64, 7, dbg);
        else ME_LAUNCH_MME3(64, 6, dbg);
#undef ME_LAUNCH_MME6
#else
        ME_LAUNCH_MME3(32, 8, 0);
#endif
#undef ME_LAUNCH_MME3
    }
    const long long chunk = ((long long) nb + kStage - 1) / kStage;
    hipLaunchKernelGGL(k_mme_final, dim3(kStage), dim3(256), 0, ctx->stream, ps, pc, (long long) nb, chunk, ps2, pc2);
    hipLaunchKernelGGL(k_mme_final, dim3(1), dim3(256), 0, ctx->stream, ps2, pc2, (long long) kStage, (long long) kStage, outs, outc);
    double hs = 0;
    long long hc = 0;
    ME_TRY(mail_post(ctx, &hs, outs, 8));  // (not hipMemcpyAsync: me_ctx::mail_h)
    ME_TRY(mail_post(ctx, &hc, outc, 8));
    if (entropies || valid) {
        DevBuf &eo = ctx->tmp[2], &vo = ctx->tmp[3];
        ME_CHECK(ctx, eo.ensure((size_t) n * 8));
        ME_CHECK(ctx, vo.ensure((size_t) n));
        if (ctx->shard_world > 1) {
            ME_CHECK(ctx, hipMemsetAsync(eo.p, 0, (size_t) n * 8, ctx->stream));
            ME_CHECK(ctx, hipMemsetAsync(vo.p, 0, (size_t) n, ctx->stream));
        }
        if (e > b)
            hipLaunchKernelGGL(k_mme_unpermute, dim3((unsigned int) ((e - b + 255) / 256)), dim3(256), 0, ctx->stream,
                               c.sp.as<SPoint>(), b, e, ent_s.as<double>(), val_s.as<unsigned char>(),
                               entropies ? eo.as<double>() : nullptr, valid ? vo.as<unsigned char>() : nullptr);
        if (entropies) ME_CHECK(ctx, hipMemcpyAsync(entropies, eo.p, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (valid) ME_CHECK(ctx, hipMemcpyAsync(valid, vo.p, (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
    }
    ME_TRY(mail_sync(ctx));
    ME_CHECK(ctx, hipGetLastError());
#ifdef ME_MME_STATS
    {
        unsigned long long st[8] = {0}, zero[8] = {0};
        (void) hipMemcpyFromSymbol(st, HIP_SYMBOL(g_mme_stat), sizeof st);
        (void) hipMemcpyToSymbol(HIP_SYMBOL(g_mme_stat), zero, sizeof zero);
        if (st[0])
            std::fprintf(stderr, "[mme stats] queries=%lld pass rounds=%llu (%.3f per 16 queries) candidates/round=%.1f queries served/round=%.1f accepted/query=%.1f | MFMA blocks/round=%.2f steps taken/round=%.2f (of %.2f) pairs/step=%.1f\n",
                         (long long) (e - b), st[0], (double) st[0] * 16.0 / (double) (e - b), (double) st[1] / (double) st[0],
                         (double) st[2] / (double) st[0], (double) st[3] / (double) (e - b) - 1.0, (double) st[4] / (double) st[0],
                         (double) st[5] / (double) st[0], 4.0 * (double) st[4] / (double) st[0],
                         st[5] ? (double) st[3] / (double) st[5] : 0.0);
    }
#endif
    c.mme_have = true;
    if (sum_H) *sum_H = hs;
    if (n_valid) *n_valid = hc;
    return ME_OK;
}

}  // namespace me
