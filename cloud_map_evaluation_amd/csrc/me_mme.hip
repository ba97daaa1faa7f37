// me_mme.hip — Mean Map Entropy: radius neighbourhood -> covariance -> 0.5*ln(2*pi*e*det)
// (ComputeMeanMapEntropyUsingNormalTBB map_eval.cpp:1608-1737; the OpenMP and serial variants :1538-1606,
//  :1438-1535 compute the same thing with k >= 10 / k >= 5).
//
// Mapping.  Points are sorted along the Hilbert curve on a grid whose cell edge is (a hair above) the search radius, so the
// neighbours of every point of a cell lie in the 3x3x3 block around it, and each cell is one contiguous run of the sorted array.
// A wavefront owns 64 consecutive sorted points.  Per round (1.2 per wave on the bench's map, 1.07 on its ground truth) it
//   * groups the lanes whose cell is within Chebyshev distance 2 of a leader's (the best of five candidates along the pending
//     stretch; normally the whole wave) and resolves the group's cell box grown by one with one hash probe per (lane, slot) into a
//     wave-private LDS run table, runs adjacent to no lane of the group culled (wave_group_table<1, true>),
//   * stages every run ONCE through a wave-private LDS tile — one coalesced vector load per 32 points, the staging lane storing an
//     FP32 record (p', |p'|^2) and u = p - o in fp64, o = the leader's point — and reads it back with broadcast ds_reads, every
//     group lane testing the candidate against its own query: an FP32 pre-test with an exact fp64 band (k_mme3 below), then
//     k, sum(u), sum(u u^T) accumulated straight from the tile (the covariance is translation-invariant; the sums start at minus
//     the lane's own contribution: the element the reference erases at :1672-1673).
// The kernel is bound by vector instruction issue (96 % busy: profiles/r04_mme3_c4_sq_per_wave.json): 57 VGPRs, eight waves per
// SIMD, no scratch; per step on the 50 M + 50 M pair (ME_MME_DBG builds, round 6) table + prologue + epilogue 3.3 ms, staging 4.7
// (hidden by the other waves in the full kernel: prefetching the next chunk made it slower), the pre-test 6.7, the accumulation —
// nine fp64 instructions for every candidate SOME lane accepts, at 12 - 17 % lane efficiency — 8.4.  What was measured and dropped
// (a per-lane walk, half-radius cells, sub-wave groups, scalar-cache delivery, readlane broadcasts, per-lane compaction, streaming
// waves, the f32 / int8 matrix pipes, per-octet candidate streams): profiles/EXPERIMENTS.md.
// The reference materialises index / distance vectors per query and gathers a 3 x k matrix; here nothing is materialised.
// Neighbourhoods so thin that the rounding of the leader-origin sums would show (smallest covariance eigenvalue below ~1.8e-6
// cell^2) are recomputed two-pass about the query itself by k_mme_refine.
#include <cmath>
#include <cstdlib>

#include "me_internal.hpp"

#ifndef ME_MME_DEPTH
#define ME_MME_DEPTH 2
#endif
#ifndef ME_MME_DBG
#define ME_MME_DBG 0  // measurement builds of k_mme3 (profiles/EXPERIMENTS.md): 1 no candidate streaming, 2 pre-test only (nothing accepted), 3 staging only (no test loop)
#endif

namespace me {

// sum over the two 32-lane halves of a wave (lanes l and l ^ 32 end with the total, bit-identical)
__device__ __forceinline__ double half2_sum_d(double v) {
    unsigned int lo = (unsigned int) __double2loint(v), hi = (unsigned int) __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int) d[0], (int) c[0]) + __hiloint2double((int) d[1], (int) c[1]);
}


#ifdef ME_MME_STATS
// build with -DME_MME_STATS (profiles/README.md): per launch, [0] wave rounds, [1] candidates streamed, [2] lanes served,
// [3] accepted (query, candidate) pairs, [4..6] rounds / candidates / lanes served of the rounds after a wave's first —
// printed to stderr by mme_run
__device__ unsigned long long g_mme_stat[8];
#endif

// one staged candidate of k_mme3's wave-private LDS tile (48 bytes: every member 16-byte aligned for ds_read_b128)
struct alignas(16) MmeTileRec {
    float4 f;    // (p'x, p'y, p'z, |p'|^2) in FP32, p' = p - o
    double2 xy;  // u = p - o in fp64
    double z;
    double pad;
};
static_assert(sizeof(MmeTileRec) == 48, "tile record layout");
// LDS load through a 32-bit LDS byte address held in a vector register
template <class T>
__device__ __forceinline__ T lds_load(unsigned int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const __attribute__((address_space(3))) T *>((size_t) addr);
#else
    (void) addr;  // (host pass of the single-source compile: never called)
    return T{};
#endif
}

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
       unsigned int xcd_chunk, int dbg, double cell_h, float thr_lo, float thr_hi, unsigned long long *__restrict__ part_pairs,
       unsigned int *__restrict__ refine_list, unsigned int *__restrict__ refine_count, double cond) {
    // refine_list / refine_count / cond (round 6): lanes whose covariance is so thin that the rounding of the leader-origin sums shows
    // (det < (tr / 2)^2 * cond, i.e. smallest eigenvalue below ~cond = 1.8e-6 h^2: a sheet thinner than ~0.2 mm at h = 0.1 m) are
    // appended to refine_list; k_mme_refine recomputes them with a two-pass covariance about the query itself (mme_run).
    // part_pairs (instrumentation, NULL in product runs): per block, the accepted (query, neighbour) pairs — what bench.py's
    // roofline.valu divides by the fp64 vector peak
    // cell_h = edge of a radius-grid cell; thr_lo / thr_hi = r^2 -+ E in FP32 (E = 2^-12 cell_h^2): kernel arguments, i.e.
    // scalar registers for the whole kernel (computed in the kernel they lived in VGPRs and were spilled around the loop).
    // dbg: profiling switches (profiles/README.md) — 1: no candidate streaming at all, 2: pre-test only, nothing accepted, 3: staging only.
    static_assert(TILE * 16 >= kGroupRows * 4, "the row masks of the cull alias the tile");
    (void) fr;
    (void) cell_h;
    const unsigned int vb = xcd_virtual_block(blockIdx.x, gridDim.x, xcd_chunk);
    const unsigned int loc = vb * blockDim.x + threadIdx.x;
    bool active = i_begin + (long long) loc < i_end;
    const int shift3 = 3 * g.shift;
    const int cell_lim = 1 << (kMortonBits - g.shift);

    // The query point is NOT kept in registers across the kernel (round 5): a lane needs it when its round starts (origin, FP32
    // thresholds, the self term) and, a few times in a thousand candidates, for the exact band test — it is loaded there (one coalesced
    // 32-byte read per lane and round, an L2 hit after the first).  Six registers fewer alive in the candidate loop: round 5's first
    // leader-origin version kept it and spilled 20 bytes per lane and round — 2.5 GB of scratch traffic per 50 M-query launch
    // (rocprofv3 FETCH + WRITE 6.3 GB against 3.8 before; profiles/EXPERIMENTS.md).
    unsigned long long mycell = ~0ULL;
    if (active) {
        const long long i = i_begin + (long long) loc;
        mycell = codes[i] >> shift3;
        if (slab.axis >= 0) {
            const SPoint q = sp[i];
            if (!slab_owned(slab, q.x, q.y, q.z)) {
                ent_s[i] = 0.0;
                valid_s[i] = 0;
                active = false;
            }
        }
    }
    bool done = !active;

    __shared__ int2 s_tab[4][kGroupTab + 1];
    // One 48-byte record per staged candidate — the FP32 record, (ux, uy), uz — read back with a VECTOR-register address and
    // immediate offsets (round 5): with three arrays indexed by the (scalar) loop counter the compiler rebuilt every LDS address
    // with a v_mov from a scalar register, three vector instructions per accepted candidate that did no arithmetic.
    __shared__ MmeTileRec s_rec[4][TILE];
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));  // scalar: the wave's LDS bases stay out of VGPRs
    int2 *tab = s_tab[wv];
    MmeTileRec *trec = s_rec[wv];
    const unsigned int trec_lds = (unsigned int) (size_t) trec;  // LDS byte address of the wave's tile (flat -> local: the low 32 bits)
    const int lane = threadIdx.x & 63;

    unsigned int wave_pairs = 0;  // (scalar: instrumentation only)
    double det_keep = 0.0;  // determinant of the neighbourhood covariance (valid when have_det)
    bool have_det = false;  // the query has at least min_k neighbours
    bool thin = false;      // ... and its covariance is thin enough for k_mme_refine
    // A lane accumulates in exactly ONE round (the one whose group it belongs to), so the moments live inside the round:
    // nothing of them is alive while the next round's table is built.
#ifdef ME_MME_STATS
    int st_round = 0;
#endif
    while (__ballot(!done)) {
        GroupBox bx;
        int nk = 0, leader = 0;
        const int cx = (int) compact21(mycell), cy = (int) compact21(mycell >> 1), cz = (int) compact21(mycell >> 2);
        const bool in = wave_group_table<1, true>(!done, cx, cy, cz, g, cell_lim, lane, tab, bx, &nk, reinterpret_cast<unsigned int *>(trec),
                                                  nullptr, nullptr, nullptr, &leader);
        // Round origin o = the group LEADER'S POINT, wave-uniform in scalar registers.  Both the FP32 record and the fp64
        // coordinates of a staged candidate are taken relative to it ONCE, by the lane that loads the candidate: u = p - o.  The
        // covariance is translation-invariant, so a lane accumulates sum(u), sum(u u^T) of what it accepts as they come out of the
        // tile — 3 v_add_f64 + 6 v_fma_f64 per accepted candidate instead of 12 fp64 instructions (no per-lane p - q) — and moves
        // nothing: cov = (S2 - S1 S1^T / k) / (k - 1) about any origin.  |u| < 4 cells (leader's cell +- 2 for the group, + 1 for the
        // stencil, + the offset inside a cell), well inside the 7-cell box the FP32 error bound E was derived for; the cancellation
        // in S2 - S1 S1^T / k costs <= |u|^2 / var ~ 10^2..10^3 of fp64's 10^16: entropies move by ~1e-13.  An exactly degenerate
        // neighbourhood that is axis-aligned (a floor z = const, a line, duplicates of the leader) still sums exact zeros.
        double qx = 0, qy = 0, qz = 0;
        if (!done) {
            unsigned int loc_q = loc;
            asm volatile("" : "+v"(loc_q));  // (the 64-bit index is rebuilt here, not carried across the rounds)
            const SPoint q = sp[i_begin + (long long) loc_q];
            qx = q.x;
            qy = q.y;
            qz = q.z;
        }
        const double ox = uniform_f64(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(qx), leader), __builtin_amdgcn_readlane(__double2loint(qx), leader)));
        const double oy = uniform_f64(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(qy), leader), __builtin_amdgcn_readlane(__double2loint(qy), leader)));
        const double oz = uniform_f64(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(qz), leader), __builtin_amdgcn_readlane(__double2loint(qz), leader)));
        const double rx = qx - ox, ry = qy - oy, rz = qz - oz;  // the lane's own u (the value its staging lane will store for it)
        const float ax = (float) (-2.0 * rx), ay = (float) (-2.0 * ry), az = (float) (-2.0 * rz);
        const float s = fmaf(az, az, fmaf(ay, ay, ax * ax));
        // (the group predicate rides on the thresholds: lanes outside the group accept nothing)
        const float t_hi = (in && dbg != 2) ? fmaf(-0.25f, s, thr_hi) : -INFINITY;
        const float t_lo = (in && dbg != 2) ? fmaf(-0.25f, s, thr_lo) : -INFINITY;
        // The query itself is one of the candidates (d2 = 0 < r^2: always accepted, the element the reference erases,
        // map_eval.cpp:1672-1673): the sums START at minus its contribution, so that nothing of q is needed after this point.
        int k = -1;
        double s1x = -rx, s1y = -ry, s1z = -rz;
        double sxx = -(rx * rx), sxy = -(rx * ry), sxz = -(rx * rz), syy = -(ry * ry), syz = -(ry * rz), szz = -(rz * rz);
        if (!in || dbg == 2) {  // (a lane outside the group accepts nothing, its own point included)
            k = 0;
            s1x = s1y = s1z = sxx = sxy = sxz = syy = syz = szz = 0.0;
        }
#ifdef ME_MME_STATS
        int st_cand = 0;
#endif
        auto test = [&](const float4 &c, unsigned int ra, int gj) {  // ra: LDS address of the tile record, gj: the candidate's position in `sp`
            const float u = fmaf(c.x, ax, fmaf(c.y, ay, fmaf(c.z, az, c.w)));
            const bool hi = u < t_hi;
            const unsigned long long mh = __ballot(hi);
            if (mh) {  // some lane may hold this candidate inside its radius
                bool acc = u < t_lo;
                // (both compares land in scalar register pairs: the band test is scalar work, no VALU instruction)
                if (__builtin_expect(mh != __ballot(acc), 0)) {  // a lane in the band: its exact test decides
                    asm volatile("; band: exact test" ::: "memory");     // (keeps the compiler from evaluating it always)
                    // (the tile holds u = p - o, the registers no q: the exact test takes both points from memory — a few
                    // candidates in a thousand)
                    unsigned int loc_b = loc;
                    asm volatile("" : "+v"(loc_b));
                    const SPoint qe = sp[i_begin + (long long) loc_b];
                    const SPoint pe = sp[gj];
                    const double ex = pe.x - qe.x, ey = pe.y - qe.y, ez = pe.z - qe.z;
                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                    acc = acc || (hi && d2 < r2);                        // strict, nanoflann RadiusResultSet [upstream]
                }
                if (acc) {
                    const double2 pxy = lds_load<double2>(ra + 16);
                    const double dx = pxy.x, dy = pxy.y, dz = lds_load<double>(ra + 32);  // u = p - o, staged
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
                    trec[lane].f = make_float4(fx, fy, fz, (float) w);
                    trec[lane].xy = make_double2(px, py);
                    trec[lane].z = pz;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int j = (dbg == 3) ? n : 0;
                // (two records in flight, not four: eight registers fewer is what keeps this kernel at 64 VGPRs without
                // spilling inside the run loop — the spills of the four-deep version were 3.2x the kernel's useful HBM traffic)
                unsigned int ra = trec_lds;
                asm volatile("" : "+v"(ra));  // the record address lives in a VECTOR register: reads below use it + an immediate
                for (; j + 2 <= n; j += 2) {
                    const float4 c0 = lds_load<float4>(ra), c1 = lds_load<float4>(ra + (unsigned int) sizeof(MmeTileRec));
                    test(c0, ra, base + j);
                    test(c1, ra + (unsigned int) sizeof(MmeTileRec), base + j + 1);
                    ra += 2u * (unsigned int) sizeof(MmeTileRec);
                    asm volatile("" : "+v"(ra));
                }
                for (; j < n; ++j) {
                    const float4 c0 = lds_load<float4>(ra);
                    test(c0, ra, base + j);
                    ra += (unsigned int) sizeof(MmeTileRec);
                    asm volatile("" : "+v"(ra));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the tile is overwritten by the next chunk
            }
        });
#ifdef ME_MME_STATS
        {
            int ka = in ? k : 0;
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
        if (part_pairs) {  // (wave-uniform branch)
            int ka = in ? k : 0;
            for (int o = 32; o > 0; o >>= 1) ka += __shfl_xor(ka, o, 64);
            wave_pairs += (unsigned int) __builtin_amdgcn_readfirstlane(ka);
        }
        if (in) {
            done = true;
            const int kk = k;  // neighbours without the query itself (map_eval.cpp:1672-1673): k started at -1
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
                const double tr = (cxx + cyy) + czz;
                thin = det_keep < 0.25 * tr * tr * cond;  // lambda_3 >= det / (lambda_1 lambda_2) >= 4 det / tr^2
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (refine_list) {  // wave-aggregated append (any order: every listed query is recomputed on its own)
        const unsigned long long tm = __ballot(thin);
        if (tm) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(refine_count, (unsigned int) __popcll(tm));
            base = (unsigned int) readlane_i((int) base, 0);
            if (thin) refine_list[base + (unsigned int) __popcll(tm & ((1ULL << lane) - 1ULL))] = loc;
        }
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
    if (part_pairs) {
        __shared__ unsigned int smp[4];
        if (lane == 0) smp[wv] = wave_pairs;
        __syncthreads();
        if (threadIdx.x == 0) part_pairs[blockIdx.x] = (unsigned long long) smp[0] + smp[1] + smp[2] + smp[3];
    }
}


// ------------------------------------------------------------------------------------------------------------
// k_mme_refine (round 6) — the thin neighbourhoods again, two-pass and about the query itself.
// k_mme3 sums u and u u^T about the round LEADER's point (|u| up to ~4 cells): the rounding it accumulates, ~eps |u|^2 per entry,
// divided by the smallest eigenvalue of the covariance, is what its entropy is off by — 6e-11 for a surface with 1 mm of relief,
// 9e-7 at 10 um, 6e-5 at 1 um (tests/test_gpu_degenerate.py, tilted sheets against the reference's own loops).  The lanes k_mme3
// flags (smallest eigenvalue below ~1.8e-6 h^2) are recomputed here the way the reference does it (map_eval.cpp:1684-1689) — mean
// first, then the centred products — but with every offset taken from the QUERY (d = p - q, exact for neighbours), so that no sum
// ever holds a term larger than r^2.  What is left is the conditioning of the 3x3 cofactor determinant itself (eps (r^2/4)^3 / det),
// which the reference's own arithmetic has too.  Eight lanes per query: lane c probes cells c, c + 8, c + 16, c + 24 of the 3x3x3
// block, the runs are scanned 32 points per trip (four independent loads per lane), the partial sums meet in three DPP stages.
// The accepted set is the exact test's (d2 < r2, strict), the element the reference erases (the query itself, d2 = 0) is taken out
// algebraically: it contributes 0 to sum(d) and (0 - m)(0 - m)^T to the centred products.
// ------------------------------------------------------------------------------------------------------------
constexpr int kRefineBlock = 256;  // 32 queries per block
__global__ void __launch_bounds__(kRefineBlock)
k_mme_refine(const SPoint *__restrict__ sp, const unsigned long long *__restrict__ codes, long long i_begin, GridView g, double r2,
             int min_k, const unsigned int *__restrict__ list, unsigned int n_list, double *__restrict__ ent_s,
             unsigned char *__restrict__ valid_s) {
    __shared__ int2 s_runs[kRefineBlock / 8][28];  // (28: the rows start in different banks)
    const int sub = threadIdx.x & 7, oct = threadIdx.x >> 3;
    const unsigned int t = blockIdx.x * (kRefineBlock / 8) + oct;
    const bool alive = t < n_list;
    const long long i = i_begin + (long long) (alive ? list[t] : 0u);
    const SPoint q = sp[alive ? i : i_begin];
    const unsigned long long cell = codes[alive ? i : i_begin] >> (3 * g.shift);
    const int cx = (int) compact21(cell), cy = (int) compact21(cell >> 1), cz = (int) compact21(cell >> 2);
    const int cell_lim = 1 << (kMortonBits - g.shift);
    {   // the 27 runs of the block, four probes per lane in flight
        unsigned long long key[4], got[4];
        unsigned int slot[4];
        bool want[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = sub + 8 * u;
            const int ix = cx + (c % 3) - 1, iy = cy + ((c / 3) % 3) - 1, iz = cz + (c / 9) - 1;
            want[u] = alive && c < 27 && ix >= 0 && iy >= 0 && iz >= 0 && ix < cell_lim && iy < cell_lim && iz < cell_lim;
            key[u] = spread21((unsigned long long) ix) | (spread21((unsigned long long) iy) << 1) | (spread21((unsigned long long) iz) << 2);
            slot[u] = (unsigned int) hash_u64(key[u]) & g.hmask;
            got[u] = kEmptyKey;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (want[u]) got[u] = g.hkeys[slot[u]];
        unsigned int ci[4], c0[4], c1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            while (want[u] && got[u] != key[u] && got[u] != kEmptyKey) {  // (collisions: linear probing, rare)
                slot[u] = (slot[u] + 1) & g.hmask;
                got[u] = g.hkeys[slot[u]];
            }
            want[u] = want[u] && got[u] == key[u];
            ci[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (want[u]) ci[u] = g.hvals[slot[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0[u] = c1[u] = 0;
            if (want[u]) {
                c0[u] = g.cell_start[ci[u]];
                c1[u] = g.cell_start[ci[u] + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (sub + 8 * u < 27) s_runs[oct][sub + 8 * u] = make_int2((int) c0[u], (int) (c1[u] - c0[u]));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (an octet lies inside one wavefront)
    // f(d, inside) for every point of the block, d = p - q
    auto scan = [&](auto &&f) {
        for (int c = 0; c < 27; ++c) {
            const int2 run = s_runs[oct][c];
            for (int j0 = 0; j0 < run.y; j0 += 32) {
                SPoint p[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) p[u] = sp[run.x + min(j0 + sub + 8 * u, run.y - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double dx = p[u].x - q.x, dy = p[u].y - q.y, dz = p[u].z - q.z;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;  // the exact test's expression
                    if (j0 + sub + 8 * u < run.y && d2 < r2) f(dx, dy, dz);  // strict, nanoflann RadiusResultSet [upstream]
                }
            }
        }
    };
    auto oct_sum = [](double v) {
        v += octet_partner_d<0>(v);
        v += octet_partner_d<1>(v);
        v += octet_partner_d<2>(v);
        return v;
    };
    // pass 1: how many, and where their mean lies
    int cnt = 0;
    double s1x = 0, s1y = 0, s1z = 0;
    scan([&](double dx, double dy, double dz) {
        ++cnt;
        s1x += dx;
        s1y += dy;
        s1z += dz;
    });
    cnt += octet_partner_i<0>(cnt);
    cnt += octet_partner_i<1>(cnt);
    cnt += octet_partner_i<2>(cnt);
    s1x = oct_sum(s1x);
    s1y = oct_sum(s1y);
    s1z = oct_sum(s1z);
    const int kk = cnt - 1;  // without the query itself (map_eval.cpp:1672-1673)
    double H = 0.0;
    bool ok = false;
    if (kk >= min_k) {  // (octet-uniform)
        const double mx = s1x / (double) kk, my = s1y / (double) kk, mz = s1z / (double) kk;  // rowwise().mean() (:1684), relative to q
        // pass 2: the centred products (:1685-1688)
        double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
        scan([&](double dx, double dy, double dz) {
            const double ex = dx - mx, ey = dy - my, ez = dz - mz;
            sxx = fma(ex, ex, sxx);
            sxy = fma(ex, ey, sxy);
            sxz = fma(ex, ez, sxz);
            syy = fma(ey, ey, syy);
            syz = fma(ey, ez, syz);
            szz = fma(ez, ez, szz);
        });
        sxx = oct_sum(sxx);
        sxy = oct_sum(sxy);
        sxz = oct_sum(sxz);
        syy = oct_sum(syy);
        syz = oct_sum(syz);
        szz = oct_sum(szz);
        // the erased element: d = 0
        sxx -= mx * mx;
        sxy -= mx * my;
        sxz -= mx * mz;
        syy -= my * my;
        syz -= my * mz;
        szz -= mz * mz;
        const double inv = 1.0 / (double) (kk - 1);
        const double cxx = sxx * inv, cxy = sxy * inv, cxz = sxz * inv, cyy = syy * inv, cyz = syz * inv, czz = szz * inv;
        const double det = cxx * (cyy * czz - cyz * cyz) - cxy * (cxy * czz - cyz * cxz) + cxz * (cxy * cyz - cyy * cxz);
        const double h = 0.5 * log(2.0 * M_PI * M_E * det);  // ComputeEntropy (:1656)
        if (!isnan(h) && !isinf(h)) {                         // (:1692)
            H = h;
            ok = true;
        }
    }
    if (alive && sub == 0) {
        ent_s[i] = H;
        valid_s[i] = ok ? 1 : 0;
    }
}

// the block partials of k_mme3 again, from the per-point arrays (after k_mme_refine has rewritten some of them): block b sums the 256
// sorted points from i_begin + 256 b on — any fixed order is a deterministic sum
__global__ void __launch_bounds__(256)
k_mme_resum(const double *__restrict__ ent_s, const unsigned char *__restrict__ valid_s, long long i_begin, long long i_end,
            double *__restrict__ part_sum, long long *__restrict__ part_cnt) {
    const long long i = i_begin + (long long) blockIdx.x * 256 + threadIdx.x;
    double H = 0.0;
    long long ok = 0;
    if (i < i_end) {
        H = ent_s[i];
        ok = valid_s[i] ? 1 : 0;
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
    const double bs = block_sum_256(H, smd);
    const long long bc = block_sum_256_ll(ok, smi);
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = bs;
        part_cnt[blockIdx.x] = bc;
    }
}

__global__ void __launch_bounds__(256) k_sum_u64(const unsigned long long *__restrict__ v, long long n, unsigned long long *__restrict__ out) {
    long long s = 0;
    for (long long i = threadIdx.x; i < n; i += 256) s += (long long) v[i];
    __shared__ long long sm[4];
    const long long r = block_sum_256_ll(s, sm);
    if (threadIdx.x == 0) *out += (unsigned long long) r;
}

#ifdef ME_AB  // measurement build only (make -C profiles/ab): the variants that were measured and not adopted live in profiles/ab/
#include "me_mme_kernels_ab.inc"
#endif

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

// per-point results of the sorted range [b, e) back in cloud order, copied to the host (asynchronous on the context's stream)
static int mme_unpermute_to_host(me_ctx *ctx, Cloud &c, long long b, long long e, double *entropies, uint8_t *valid) {
    const long long n = c.n;
    DevBuf &eo = ctx->tmp[2], &vo = ctx->tmp[3];
    ME_CHECK(ctx, eo.ensure((size_t) n * 8));
    ME_CHECK(ctx, vo.ensure((size_t) n));
    if (ctx->shard_world > 1) {
        ME_CHECK(ctx, hipMemsetAsync(eo.p, 0, (size_t) n * 8, ctx->stream));
        ME_CHECK(ctx, hipMemsetAsync(vo.p, 0, (size_t) n, ctx->stream));
    }
    if (e > b)
        hipLaunchKernelGGL(k_mme_unpermute, dim3((unsigned int) ((e - b + 255) / 256)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), b, e,
                           c.mme_ent.as<double>(), c.mme_val.as<unsigned char>(), entropies ? eo.as<double>() : nullptr,
                           valid ? vo.as<unsigned char>() : nullptr);
    if (entropies) ME_TRY(copy_d2h(ctx, entropies, eo.p, (size_t) n * 8));
    if (valid) ME_TRY(copy_d2h(ctx, valid, vo.p, (size_t) n));
    return ME_OK;
}

// The per-point MME result lives in SORTED order and an index rebuild discards it (cloud_build_index).  me_run_suite_from evaluates
// the map as loaded (map_eval.cpp:56) and transforms it afterwards (:1206): mme_carry_out parks the result in cloud order before
// the transform, mme_carry_in puts it back in the new sorted order, so that me_mme_fetch / me_render_entropy still serve it.
__global__ void k_mme_repermute(const SPoint *__restrict__ sp, long long n, const double *__restrict__ ent_o,
                                const unsigned char *__restrict__ valid_o, double *__restrict__ ent_s, unsigned char *__restrict__ valid_s) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long o = sp[i].idx;
    ent_s[i] = ent_o[o];
    valid_s[i] = valid_o[o];
}

int mme_carry_out(me_ctx *ctx, int slot) {
    Cloud &c = ctx->cloud[slot];
    if (!c.mme_have || c.n == 0) return ME_OK;
    ME_CHECK(ctx, ctx->mme_keep_e.ensure((size_t) c.n * 8));
    ME_CHECK(ctx, ctx->mme_keep_v.ensure((size_t) c.n));
    hipLaunchKernelGGL(k_mme_unpermute, dim3((unsigned int) ((c.n + 255) / 256)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), 0LL, c.n,
                       c.mme_ent.as<double>(), c.mme_val.as<unsigned char>(), ctx->mme_keep_e.as<double>(),
                       ctx->mme_keep_v.as<unsigned char>());
    ctx->mme_keep_n = c.n;
    return ME_OK;
}

int mme_carry_in(me_ctx *ctx, int slot) {
    Cloud &c = ctx->cloud[slot];
    if (ctx->mme_keep_n != c.n || c.n == 0 || !c.index_valid) return ME_OK;
    ME_CHECK(ctx, c.mme_ent.ensure((size_t) c.n * 8));
    ME_CHECK(ctx, c.mme_val.ensure((size_t) c.n));
    hipLaunchKernelGGL(k_mme_repermute, dim3((unsigned int) ((c.n + 255) / 256)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), c.n,
                       ctx->mme_keep_e.as<double>(), ctx->mme_keep_v.as<unsigned char>(), c.mme_ent.as<double>(),
                       c.mme_val.as<unsigned char>());
    c.mme_have = true;
    ctx->mme_keep_n = -1;
    return ME_OK;
}

// me_mme_fetch: the per-point arrays of the last me_mme / me_run_suite* of this slot, as me_mme would have returned them
int mme_fetch(me_ctx *ctx, int slot, double *entropies, uint8_t *valid) {
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded || !c.mme_have) return ctx->fail(ME_ERR_STATE, "me_mme_fetch: no MME result for this slot (run me_mme or me_run_suite first)");
    if (c.n == 0 || (!entropies && !valid)) return ME_OK;
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    long long b, e;
    ctx->shard_range(c.n, b, e);
    ME_TRY(mme_unpermute_to_host(ctx, c, b, e, entropies, valid));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
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
    ME_TRACE_POINT(ctx, "mme_run: enter");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    // the 27-cell stencil is exact only when cell edge >= radius; rebuild when it is not, or when the cells are
    // needlessly coarse (more candidates per query than necessary)
    const double want_h = radius * (1.0 + 0x1p-20);
    if (!c.index_valid || c.cell_h < want_h || c.cell_h > 1.5 * want_h) ME_TRY(cloud_build_index(ctx, slot, radius));
    const long long n = c.n;
    long long b, e;
    ctx->shard_range(n, b, e);
#ifdef ME_AB
#include "me_mme_prepare_ab.inc"
#endif
    DevBuf &ent_s = c.mme_ent, &val_s = c.mme_val;  // kept for me_render_entropy
    c.mme_have = false;
    ME_CHECK(ctx, ent_s.ensure((size_t) n * 8));
    ME_CHECK(ctx, val_s.ensure((size_t) n));
    const unsigned int nb = (unsigned int) std::max<long long>(8, ((e - b + 255) / 256 + 7) / 8 * 8);
    constexpr int kStage = 256;
    ME_TRACE_POINT(ctx, "mme_run: buffers ensured");
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
        // instrumentation (timers on): accepted pairs per block -> ctx->mme_pairs (me_timer_get "mme_pairs")
        unsigned long long *d_pairs = nullptr;
#ifndef ME_NO_PAIR_COUNT
        if (ctx->timers_on) {
            ME_CHECK(ctx, ctx->mme_pairs_buf.ensure((size_t) (nb + 2) * 8));
            d_pairs = ctx->mme_pairs_buf.as<unsigned long long>() + 2;
        }
#endif
        // thin neighbourhoods (k_mme_refine): [0] the list's length, [1 ..] the sorted offsets of the flagged queries
        ME_CHECK(ctx, ctx->mme_refine.ensure((size_t) (e - b + 1) * 4 + 64));
        unsigned int *d_refine = ctx->mme_refine.as<unsigned int>();
        ME_CHECK(ctx, hipMemsetAsync(d_refine, 0, 4, ctx->stream));
        const double cond = ME_TUNE_MME_REFINE_COND * c.cell_h * c.cell_h;
        TimerScope ts(ctx, "mme");
#define ME_LAUNCH_MME3(T, W, DBG)                                                                                             \
    hipLaunchKernelGGL((k_mme3<T, W>), dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(),                                \
                       c.codes.as<unsigned long long>(), b, e, c.grid, fr, c.slab, r2, min_k, ent_s.as<double>(),             \
                       val_s.as<unsigned char>(), ps, pc, xcd_chunk_setting(), DBG, c.cell_h, thr_lo, thr_hi, d_pairs, d_refine + 1,      \
                       d_refine, cond)
#ifdef ME_AB
#include "me_mme_dispatch_ab.inc"
#else
        ME_LAUNCH_MME3(32, ME_TUNE_MME_WAVES, ME_MME_DBG);
#endif
#undef ME_LAUNCH_MME3
        ts.end();
        ME_TRACE_POINT(ctx, "mme_run: k_mme3 queued");
        if (d_pairs) {
            unsigned long long *tot = ctx->mme_pairs_buf.as<unsigned long long>();
            ME_CHECK(ctx, hipMemsetAsync(tot, 0, 8, ctx->stream));
            hipLaunchKernelGGL(k_sum_u64, dim3(1), dim3(256), 0, ctx->stream, d_pairs, (long long) nb, tot);
            unsigned long long h = 0;
            {
                MailGuard mg(ctx);
                ME_TRY(mail_post(ctx, &h, tot, 8));
                ME_TRY(mg.sync());
            }
            ctx->mme_pairs += (long long) h;
        }
    }
    ME_TRACE_POINT(ctx, "mme_run: kernel launched");
    const long long chunk = ((long long) nb + kStage - 1) / kStage;
    hipLaunchKernelGGL(k_mme_final, dim3(kStage), dim3(256), 0, ctx->stream, ps, pc, (long long) nb, chunk, ps2, pc2);
    hipLaunchKernelGGL(k_mme_final, dim3(1), dim3(256), 0, ctx->stream, ps2, pc2, (long long) kStage, (long long) kStage, outs, outc);
    double hs = 0;
    long long hc = 0;
    unsigned int n_thin = 0;
    {   // the two sums (and how many neighbourhoods were flagged as thin) through the mailbox (not hipMemcpyAsync: me_ctx::mail_h),
        // posted after everything above that can fail — their destinations are locals of this frame (MailGuard)
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, &hs, outs, 8));
        ME_TRY(mail_post(ctx, &hc, outc, 8));
        ME_TRY(mail_post(ctx, &n_thin, ctx->mme_refine.as<unsigned int>(), 4));
        ME_TRY(mg.sync());
    }
    if (n_thin > 0) {
        // (rare: surfaces with less than ~0.2 mm of relief inside the radius — synthetic planes, CAD samples)  The flagged queries
        // are recomputed two-pass about themselves, then the block partials are formed again from the per-point arrays.
        ME_TRACE_POINT(ctx, "mme_run: refining thin neighbourhoods");
        {
            TimerScope ts(ctx, "mme_refine");
            hipLaunchKernelGGL(k_mme_refine, dim3((n_thin + kRefineBlock / 8 - 1) / (kRefineBlock / 8)), dim3(kRefineBlock), 0, ctx->stream,
                               c.sp.as<SPoint>(), c.codes.as<unsigned long long>(), b, c.grid, r2, min_k,
                               (const unsigned int *) (ctx->mme_refine.as<unsigned int>() + 1), n_thin, ent_s.as<double>(), val_s.as<unsigned char>());
        }
        const unsigned int nb2 = (unsigned int) ((e - b + 255) / 256);  // (<= nb: the partial arrays are large enough)
        hipLaunchKernelGGL(k_mme_resum, dim3(nb2), dim3(256), 0, ctx->stream, (const double *) ent_s.as<double>(),
                           (const unsigned char *) val_s.as<unsigned char>(), b, e, ps, pc);
        const long long chunk2 = ((long long) nb2 + kStage - 1) / kStage;
        hipLaunchKernelGGL(k_mme_final, dim3(kStage), dim3(256), 0, ctx->stream, ps, pc, (long long) nb2, chunk2, ps2, pc2);
        hipLaunchKernelGGL(k_mme_final, dim3(1), dim3(256), 0, ctx->stream, ps2, pc2, (long long) kStage, (long long) kStage, outs, outc);
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, &hs, outs, 8));
        ME_TRY(mail_post(ctx, &hc, outc, 8));
        ME_TRY(mg.sync());
        ctx->mme_refined += (long long) n_thin;
    }
    if (entropies || valid) ME_TRY(mme_unpermute_to_host(ctx, c, b, e, entropies, valid));
    ME_TRACE_POINT(ctx, "mme_run: synced");
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
