// me_nn.hip — batched exact 1-NN (KDTreeFlann::SearchKNN k=1, map_eval.cpp:1218,1231,1415,1424) and the
// threshold statistics of getDiffRegResultWithCorrespondence (map_eval.cpp:1069-1145).
//
// Two kernels: a uniform-grid fast path (k_nn_grid) and, for what it cannot settle, a general walk (k_nn1) of a sparse
// octree over the Morton prefixes: eight lanes per query (one per child), stackless AND nearest-first:
//   state = (level, node, one byte of "children already taken" per level).
//   At a node the <= 8 child records (contiguous, one burst of loads) are bounded at once; the closest
//   not-yet-taken child whose lower bound does not exceed the current best is entered; when none is left the
//   walk returns to the parent (whose child bounds are simply recomputed).  Nearest-first order makes the
//   first leaf reached almost always the right one, so far-away queries (outliers, non-overlapping regions;
//   the README run has FULL CD = 102 m) stay cheap instead of sweeping every node inside a loose bound.
//   The bound is computed in fp64 from fp32 boxes rounded outward and every operation is monotone, so it
//   never exceeds the *computed* distance of a point inside: the result is the exact brute-force minimum of
//   ((dx*dx + dy*dy) + dz*dz), bit-identical to the CPU path.
#include <cmath>
#include <cstdlib>

#include "me_internal.hpp"

#ifndef ME_NN_DBG
#define ME_NN_DBG 0  // measurement builds of k_nn_grid (profiles/EXPERIMENTS.md "Round 6"): 1 no ranking loop, 2 no staging either, 3 no exact epilogue loads, 4 ranking without note()
#endif

namespace me {

// v_min_f64 without the NaN canonicalisation the compiler wraps around fmin() (squared distances are never NaN)
__device__ __forceinline__ double vmin_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// v_min_f32 without the canonicalisation (ranks are never NaN)
__device__ __forceinline__ float fminf_raw(float a, float b) {
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double box_lower_bound(const float *__restrict__ b, double qx, double qy, double qz) {
    const double dx = fmax(fmax((double) b[0] - qx, qx - (double) b[3]), 0.0);
    const double dy = fmax(fmax((double) b[1] - qy, qy - (double) b[4]), 0.0);
    const double dz = fmax(fmax((double) b[2] - qz, qz - (double) b[5]), 0.0);
    return (dx * dx + dy * dy) + dz * dz;
}

// Lower bound of the squared distance from q to the points of box b whose coordinate along `axis` lies OUTSIDE [clo, chi) — the
// interval of that axis the query's owner has already searched completely (its slab + halo, me_nn_points_covered): no point inside
// it can beat the bound that came with the query, so only the part of a box that sticks out of it counts; a box that lies inside
// gets +inf.  q itself lies inside the interval (it is owned by that rank).  Monotone operations on the outward-rounded box: still a
// lower bound.  Without this a rank DISPROVES a far outlier of its neighbour — a ball of metres around the query that reaches across
// the face — by walking every occupied cell the ball touches on its side, most of them in the strip the owner's halo covers.
__device__ __forceinline__ double box_lower_bound_cov(const float *__restrict__ b, double qx, double qy, double qz, int axis, double clo,
                                                      double chi) {
    double d[3];
    d[0] = fmax(fmax((double) b[0] - qx, qx - (double) b[3]), 0.0);
    d[1] = fmax(fmax((double) b[1] - qy, qy - (double) b[4]), 0.0);
    d[2] = fmax(fmax((double) b[2] - qz, qz - (double) b[5]), 0.0);
    const double qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
    const double lo = axis == 0 ? (double) b[0] : (axis == 1 ? (double) b[1] : (double) b[2]);
    const double hi = axis == 0 ? (double) b[3] : (axis == 1 ? (double) b[4] : (double) b[5]);
    if (qa >= clo && qa < chi) {
        const double up = hi >= chi ? fmax(lo, chi) - qa : INFINITY;   // the part at or above the interval's upper end
        const double dn = lo < clo ? qa - fmin(hi, clo) : INFINITY;    // the part below its lower end
        const double da = fmin(up, dn);
        if (axis == 0) d[0] = da;
        else if (axis == 1) d[1] = da;
        else d[2] = da;
    }
    return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
}

// squared distance to the farthest corner of the box: an upper bound on the distance to ANY point inside it
// (monotone operations on the outward-rounded box, so it is >= the computed distance of every contained point)
__device__ __forceinline__ double box_upper_bound(const float *__restrict__ b, double qx, double qy, double qz) {
    const double dx = fmax(fabs(qx - (double) b[0]), fabs(qx - (double) b[3]));
    const double dy = fmax(fabs(qy - (double) b[1]), fabs(qy - (double) b[4]));
    const double dz = fmax(fabs(qz - (double) b[2]), fabs(qz - (double) b[5]));
    return (dx * dx + dy * dy) + dz * dz;
}

// ------------------------------------------------------------------------------------------------------------
// Fast path: uniform-grid 1-NN.  A wavefront owns 64 consecutive (curve-ordered) queries.  The lanes whose
// reference-grid cell lies within Chebyshev distance 2 of the leader's form a group (normally the whole wave); the
// group's cell box grown by one is resolved with one hash probe per (lane, slot) into a wave-private LDS table, and
// every non-empty run is streamed ONCE through a wave-private LDS tile (one coalesced load per run, broadcast reads), every
// lane ranking the candidates in FP32 and settling the winner exactly in fp64 afterwards (see "Ranking" below).  The grid
// level is the finest whose occupied cells hold >= 6 points.
// A lane is RESOLVED when its best distance is below its distance to the faces of its own 3x3x3 block: every
// reference point outside the block is farther, so the minimum is the exact global minimum.  Unresolved lanes
// (no neighbour within about one cell edge: outliers, non-overlapping map regions, queries outside the reference
// bbox) are appended to a list, with their best-so-far as the initial bound, for the octree kernel below.
// Measured alternatives (rocprofv3 SQ/TCP counters, profiles/README.md): a per-lane walk of each lane's own 27 runs
// tests 4x fewer candidates but is bound by vector-L1 tag lookups (~11 distinct lines per load instruction) and
// loses; Chebyshev-1 groups serialise a wave into ~5 rounds on a fine grid and lose as well.
// ------------------------------------------------------------------------------------------------------------
// FROM_LIST (round 4, the second pass of the cascade): the queries are q_begin + in_list[t], t < *in_count — what the first pass
// on the fine grid could not settle, in curve order — and `g` is a COARSER level (the radius grid): a query whose neighbour is a few
// fine cells away (drift, noise: 4.5 % of the queries at 10^4 pts/m^2, where the fine cells are 2.5 cm) is settled here, by the same
// streamed ranking, instead of walking the octree (29 of 90 ms per step on that scene).  A fixed grid of waves strides over the list
// (its length stays on the device).  First pass: unresolved queries are FLAGGED (flag_out), a stream compaction builds the ordered list.
template <bool FROM_LIST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_nn_grid(const SPoint *__restrict__ qsp, long long q_begin, long long q_end, const SPoint *__restrict__ rsp, long long nr,
          GridView g,
          FrameView fr, SlabView slab, double *__restrict__ d2_out, int *__restrict__ idx_out,
          unsigned int *__restrict__ list, unsigned int *__restrict__ list_count, unsigned int xcd_chunk,
          const unsigned int *__restrict__ in_list, const unsigned int *__restrict__ in_count, unsigned char *__restrict__ flag_out) {
    const int lane = threadIdx.x & 63;
    const unsigned int vb = FROM_LIST ? blockIdx.x : xcd_virtual_block(blockIdx.x, gridDim.x, xcd_chunk);  // gridDim.x is a multiple of 8
    const int cell_bits = kMortonBits - g.shift;
    const int cell_lim = 1 << cell_bits;
    const double cell_h = ldexp(fr.fine_h, g.shift);
    __shared__ float4 s_tile[4][64];  // (doubles as the row masks of the adjacency cull while a round's table is built)
    __shared__ int2 s_tab[4][kGroupTab + 1];
    // (list positions and the offsets of a pass are 32-bit — a cloud holds < 2^31 points —: the 64-bit forms cost the list pass, whose
    // loop keeps them alive across a whole round, 12 bytes of scratch per lane)
    const unsigned int n_in = FROM_LIST ? *in_count : 0u;
  for (unsigned int pos = vb * blockDim.x + threadIdx.x;; pos += gridDim.x * blockDim.x) {
    unsigned int qoff;  // the query is qsp[q_begin + qoff] (the 64-bit index is formed where it is used, not carried)
    bool active;
    if (FROM_LIST) {
        if (!__ballot(pos < n_in)) break;  // wave-uniform
        active = pos < n_in;
        qoff = active ? in_list[pos] : 0u;
    } else {
        qoff = pos;
        active = q_begin + (long long) qoff < q_end;
    }

    double qx = 0, qy = 0, qz = 0;
    int mcx = 0, mcy = 0, mcz = 0;  // the query's cell in the REFERENCE cloud's grid
    bool in_grid = false;
    if (active) {
        const SPoint q = qsp[q_begin + (long long) qoff];
        qx = q.x;
        qy = q.y;
        qz = q.z;
        if (!FROM_LIST && !slab_owned(slab, qx, qy, qz)) {  // halo point: a reference for others, not a query of this rank
            d2_out[q_begin + (long long) qoff] = -1.0;                 // skip marker for the statistics kernels
            idx_out[q_begin + (long long) qoff] = -1;
            active = false;
        }
    }
    if (active) {
        const double fx = fine_coord(qx, fr.ox, fr.fine_h), fy = fine_coord(qy, fr.oy, fr.fine_h),
                     fz = fine_coord(qz, fr.oz, fr.fine_h);
        const double lim = 2097151.0;
        in_grid = fx >= 0.0 && fy >= 0.0 && fz >= 0.0 && fx <= lim && fy <= lim && fz <= lim;
        if (in_grid) {
            mcx = (int) ((unsigned int) fx >> g.shift);
            mcy = (int) ((unsigned int) fy >> g.shift);
            mcz = (int) ((unsigned int) fz >> g.shift);
        }
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;  // the three smallest group ranks seen (see below)
    int j1 = -1, j2 = -1;  // positions (in the sorted reference array) of the groups that produced b1 and b2
    bool done = !active || !in_grid;

    // Ranking.  argmin_p |p - q|^2 = argmin_p (|p|^2 - 2 p.q).  With the candidates shifted to a wave-local origin o (the
    // corner of the round's cell box, |p - o| and |q - o| < 8 cells) the rank
    //     r = fma(p'x, ax, fma(p'y, ay, fma(p'z, az, |p'|^2))),   p' = p - o,  a = -2 (q - o)
    // is accurate enough in FP32 to ORDER candidates down to ~1e-6 (cell edge 0.1 m) of squared distance: p' and |p'|^2
    // are computed in fp64 ONCE per candidate when its run is staged and stored as one float4, so a candidate costs one
    // 16-byte broadcast LDS read and 3 fp32 FMAs per lane (the fp64 variant of this loop was bound by LDS bandwidth:
    // 32 bytes per candidate broadcast to 64 lanes).  The loop keeps the two best GROUPS of <= 8 stream-consecutive
    // candidates (rank + position) and the third-best rank.  The epilogue evaluates both groups EXACTLY in fp64
    // ((dx*dx + dy*dy) + dz*dz, the CPU path's value; ties -> smallest original index, as the CPU path) — every
    // candidate outside them ranks at least b3, so the exact minimum over the two groups is the answer whenever
    // b3 - b1 exceeds twice the rank error (rank_tol).  The few lanes where it does not (about 0.05 %: three near-equal
    // neighbours, or duplicates spread over three groups) go to the octree kernel, which is exact.
    // Only the lanes of the current round's group rank its candidates (in_round: the exec mask of the ranking loop): a lane ranks
    // candidates in exactly one round (one origin) and never sees a candidate twice.
    float ax = 0, ay = 0, az = 0;
    bool in_round = false;  // this lane belongs to the current round's group
    // rank error: the cell box spans <= 7 cells per axis, so |p'| <= 12 h, |a| <= 24 h, every term of r is below ~150 h^2
    // and carries a few 2^-24 relative: E < 1.2e-4 h^2 in the worst case (typically 10x less); the check uses 2E with margin
    const float rank_tol = (float) (1e-3 * cell_h * cell_h);
    // (b1 <= b2 <= b3 are the three smallest ranks seen: the new second is the median of {b1, b2, m}, the new third the
    // median of {b2, b3, m} — one v_med3_f32 each; fminf/fmaxf chains cost three times as many instructions, most of them
    // NaN canonicalisations of values that are never NaN)
#if ME_TUNE_NN_SGPR_MASKS
    // The two compare masks go to SCALAR register pairs and the selects read them from there (VOP3 encodings, written out: the
    // compiler puts both masks through vcc).  A v_cndmask_b32 that takes its mask from vcc right after the v_cmp that wrote it
    // costs ~11 issue cycles on gfx950, 4.2 with the mask in an SGPR pair (profiles/r04_issue_rates.txt) — four of them per
    // group of four candidates; b1 needs no select at all (ranks are never NaN: a plain minimum).
    auto note = [&](float m, int j) {
        unsigned long long lt1, lt2;
        asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(lt1) : "v"(m), "v"(b1));
        asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(lt2) : "v"(m), "v"(b2));
        b3 = __builtin_amdgcn_fmed3f(b2, b3, m);
        b2 = __builtin_amdgcn_fmed3f(b1, b2, m);
        int jn;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(jn) : "v"(j2), "v"(j), "s"(lt2));   // lt2 ? j : j2
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(j2) : "v"(jn), "v"(j1), "s"(lt1));  // lt1 ? j1 : jn
        asm("v_min_f32_e32 %0, %1, %2" : "=v"(b1) : "v"(b1), "v"(m));                     // lt1 ? m : b1
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(j1) : "v"(j1), "v"(j), "s"(lt1));   // lt1 ? j : j1
    };
#else
    auto note = [&](float m, int j) {
        const bool lt1 = m < b1, lt2 = m < b2;
        b3 = __builtin_amdgcn_fmed3f(b2, b3, m);
        b2 = __builtin_amdgcn_fmed3f(b1, b2, m);
        const int jn = lt2 ? j : j2;  // (two selects, no branch)
        j2 = lt1 ? j1 : jn;
        b1 = lt1 ? m : b1;
        j1 = lt1 ? j : j1;
    };
#endif
    // Candidate delivery.  A run (the points of one cell) is copied into a wave-private LDS tile with ONE coalesced vector
    // load per 64 points and read back with wave-uniform (broadcast) ds_reads, four candidates per group.  The first
    // version fetched the candidates with scalar loads (s_load, candidate in SGPRs): the scalar cache misses on this
    // stream (rocprofv3 SQC counters: 75 % of the requests), so every group of four paid an L2 round trip that 8 waves
    // per SIMD could not hide (68 % VALU issue).
    float4 *tile = s_tile[threadIdx.x >> 6];
    double ox = 0, oy = 0, oz = 0;  // the round's local origin (wave-uniform)
    // A run is padded to a multiple of four with records of rank +inf: no scalar tail loop (with 6-point cells most runs end
    // in a partial group, and a lone candidate costs a whole note()).
    // (Measured and rejected: candidates stored in pairs and ranked two at a time with v_pk_fma_f32 — 19 instead of 25
    // instructions per group of four, yet 13.8 ms against 13.0: the packed FMA does not issue faster than two plain ones.)
    auto rank = [&](int j) {
        const float4 c = tile[j];
        return fmaf(c.x, ax, fmaf(c.y, ay, fmaf(c.z, az, c.w)));
    };
    // (ranks are never NaN: min3 / med3 rather than fminf, which canonicalises its operands first)
    auto min3 = [](float a, float b, float c) { return __builtin_amdgcn_fmed3f(-INFINITY, a, __builtin_amdgcn_fmed3f(-INFINITY, b, c)); };
    // One staged chunk (<= 64 points of one run, or of several runs that are contiguous in the sorted array): the lane's point of the
    // chunk arrives in (px64, py64, pz64) — loaded ONE CHUNK AHEAD (round 6, below) — is shifted to the round's origin, written to
    // the tile as an FP32 record and ranked by every lane of the group.
    double px64 = 0, py64 = 0, pz64 = 0;
    auto load_chunk = [&](int base, int n) {
        if (lane < n && ME_NN_DBG != 2) {
            const SPoint p = rsp[(ME_NN_DBG == 5) ? ((base + lane) & 4095) : (base + lane)];  // (5: staging from a cache-resident stretch)
            px64 = p.x;
            py64 = p.y;
            pz64 = p.z;
        }
    };
    auto stage_chunk = [&](int n) {
        const int n4 = (n + 3) & ~3;
        if (lane < n4 && ME_NN_DBG != 2) {
            float4 rec = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
            if (lane < n) {
                const double px = px64 - ox, py = py64 - oy, pz = pz64 - oz;
                rec = make_float4((float) px, (float) py, (float) pz, (float) fma(pz, pz, fma(py, py, px * px)));
            }
            tile[lane] = rec;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto rank_chunk = [&](int base, int n) {
        const int n4 = (n + 3) & ~3;
        // Groups of EIGHT stream-consecutive candidates per note() (round 6; four through round 5): the three-smallest update
        // and its position selects cost 11 vector instructions whatever the group holds — 25 per four candidates, 37 per eight —
        // and the epilogue pays with eight exact fp64 evaluations per kept group instead of four.  A run's last four (n4 % 8)
        // are noted as a group of their own; the epilogue evaluates eight positions from any group's start (positions past a
        // group belong to the next cell along the curve: real reference points all the same).
        // The group predicate is the EXEC mask of the ranking loop (lanes outside the round's group rank nothing), not a +inf
        // bias added to every group's minimum.
        if (in_round && ME_NN_DBG != 1 && ME_NN_DBG != 2) {
            int j = 0;
#if ME_NN_DBG == 4
            float acc = 0.0f;
            for (; j < n4; ++j) acc += rank(j);
            if (acc == 12345.0f) note(acc, base);
            j = n4;
#endif
            for (; j + 8 <= n4; j += 8) {
                const float m0 = min3(rank(j), rank(j + 1), rank(j + 2));
                const float m1 = min3(rank(j + 3), rank(j + 4), rank(j + 5));
                note(min3(m0, m1, fminf_raw(rank(j + 6), rank(j + 7))), base + j);
            }
            if (j < n4) note(min3(rank(j), rank(j + 1), fminf_raw(rank(j + 2), rank(j + 3))), base + j);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the tile is overwritten by the next chunk
    };
    // (the candidate set of a round is a superset of the lane's own 3x3x3 block; extra candidates only help)
    int2 *tab = s_tab[threadIdx.x >> 6];
    while (__ballot(!done)) {
        GroupBox bx;
        int nk = 0;
        // (cull: a run adjacent to no lane of the group is in nobody's 3x3x3 block, so nobody's resolution test needs it)
        const bool in = wave_group_table<1, true>(!done, mcx, mcy, mcz, g, cell_lim, lane, tab, bx, &nk,
                                                  reinterpret_cast<unsigned int *>(tile));
        ox = uniform_f64(fr.ox + (double) bx.x0 * cell_h);  // (scalar registers: wave-uniform)
        oy = uniform_f64(fr.oy + (double) bx.y0 * cell_h);
        oz = uniform_f64(fr.oz + (double) bx.z0 * cell_h);
        if (in) {
            // (the query point is NOT carried through the ranking loop — six vector registers the prefetched chunk point needs: it is
            // loaded again where it is used, here and in the epilogue; an L2 hit)
            unsigned int qo = qoff;
            asm volatile("" : "+v"(qo));
            const SPoint q = qsp[q_begin + (long long) qo];
            ax = (float) (-2.0 * (q.x - ox));
            ay = (float) (-2.0 * (q.y - oy));
            az = (float) (-2.0 * (q.z - oz));
        }
        in_round = in;
        // The chunks of the round, from an explicit iterator over the run table (wave-uniform state: scalar registers) with the NEXT
        // chunk's load in flight while the current one is ranked (round 6): with the load issued after the previous chunk's ranking
        // the wave sat out an L2 / HBM round trip ~20 times per round — 1.4 of 12.2 ms per step (ME_NN_DBG=5: the same kernel staging
        // from a cache-resident stretch).  Runs that are contiguous in the sorted array (the curve visits cell x + 1 right after cell
        // x) are streamed as one, as wave_for_each_run<MERGE> did.
        int it_blk = -64, it_pos = 0, it_end = 0;
        unsigned long long it_m = 0;
        auto next_chunk = [&](int &cbase, int &cn) -> bool {
            if (it_pos >= it_end) {
                while (!it_m) {
                    it_blk += 64;
                    if (it_blk >= nk) return false;
                    const int t = it_blk + lane;
                    const int cnt = (t < nk) ? tab[t].y : 0;
                    it_m = __ballot(cnt > 0);
                }
                const int sidx = __ffsll((long long) it_m) - 1;
                it_m &= it_m - 1;
                const int2 run = tab[it_blk + sidx];
                it_pos = __builtin_amdgcn_readfirstlane(run.x);
                it_end = it_pos + __builtin_amdgcn_readfirstlane(run.y);
                while (it_m) {  // contiguous successors inside this block of the table
                    const int s2 = __ffsll((long long) it_m) - 1;
                    const int2 nx = tab[it_blk + s2];
                    if (__builtin_amdgcn_readfirstlane(nx.x) != it_end) break;
                    it_end += __builtin_amdgcn_readfirstlane(nx.y);
                    it_m &= it_m - 1;
                }
            }
            cbase = it_pos;
            cn = min(64, it_end - it_pos);
            it_pos += cn;
            return true;
        };
        int cb = 0, cn = 0;
        bool have = next_chunk(cb, cn);
        if (have) load_chunk(cb, cn);
        while (have) {
            stage_chunk(cn);
            int nb2 = 0, nn2 = 0;
            const bool nhave = next_chunk(nb2, nn2);
            if (nhave) load_chunk(nb2, nn2);  // lands under the ranking below
            rank_chunk(cb, cn);
            have = nhave;
            cb = nb2;
            cn = nn2;
        }
        if (in) done = true;
        __builtin_amdgcn_wave_barrier();
    }

    bool unresolved = false;
    if (active) {
        unresolved = true;
        {
            unsigned int qo = qoff;
            asm volatile("" : "+v"(qo));
            const SPoint q = qsp[q_begin + (long long) qo];
            qx = q.x;
            qy = q.y;
            qz = q.z;
        }
        // exact distances of the two best groups (positions past a group's run belong to cells outside the candidate set:
        // real reference points all the same, so a closer one among them is a better answer, not an error)
        double best_x = INFINITY;
        long long best_i = 0x7fffffffffffffffLL;
        auto exact_group = [&](int jg) {
            if (jg < 0 || ME_NN_DBG == 3) return;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const long long pos = (long long) jg + t;
                if (pos < nr) {
                    const SPoint p = rsp[pos];
                    const double e = dist2_exact(qx, qy, qz, p.x, p.y, p.z);
                    if (e < best_x || (e == best_x && p.idx < best_i)) {
                        best_x = e;
                        best_i = p.idx;
                    }
                }
            }
        };
        exact_group(j1);
        // The second group is evaluated only where it can matter (round 6): when b2 - b1 exceeds the rank tolerance every candidate
        // outside the best group — the second group's included — is strictly farther than the best group's minimum, in exact
        // arithmetic too.  Eight scattered 32-byte loads per lane less for the large majority of the lanes (the vector memory
        // pipe's cost of a load follows its active lanes): with groups of eight the epilogue moved 25 GB per 50 M-query launch
        // through L1 / L2 and ate what the ranking loop had saved.
        if (!(b2 - b1 > rank_tol)) exact_group(j2);
        const bool ranking_safe = b3 - b1 > rank_tol;
        if (in_grid && j1 >= 0 && ranking_safe) {
            // distance from q to the faces of the 3x3x3 cell block around its cell (>= one cell edge... minus where
            // q sits in its cell); anything outside the block is at least that far away
            const double lox = fr.ox + (double) (mcx - 1) * cell_h, hix = fr.ox + (double) (mcx + 2) * cell_h;
            const double loy = fr.oy + (double) (mcy - 1) * cell_h, hiy = fr.oy + (double) (mcy + 2) * cell_h;
            const double loz = fr.oz + (double) (mcz - 1) * cell_h, hiz = fr.oz + (double) (mcz + 2) * cell_h;
            double gmin = fmin(fmin(qx - lox, hix - qx), fmin(fmin(qy - loy, hiy - qy), fmin(qz - loz, hiz - qz)));
            gmin *= (1.0 - 1e-9);  // rounding slack of the cell assignment
            unresolved = !(gmin > 0.0 && best_x < gmin * gmin);
        }
        d2_out[q_begin + (long long) qoff] = best_x;  // final if resolved, initial bound otherwise
        idx_out[q_begin + (long long) qoff] = (j1 >= 0) ? (int) best_i : -1;
    }
    if (ME_NN_DBG) unresolved = false;  // (measurement builds: nothing goes to the octree pass)
    if (!FROM_LIST && flag_out) {
        // first pass of the cascade: flag the unresolved queries; the ordered list is built by a stream compaction
        if (q_begin + (long long) qoff < q_end) flag_out[qoff] = unresolved ? 1 : 0;
    } else {
        // wave-aggregated append of the unresolved lanes
        const unsigned long long um = __ballot(unresolved);
        if (um) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(list_count, (unsigned int) __popcll(um));
            base = (unsigned int) readlane_i((int) base, 0);
            if (unresolved) list[base + (unsigned int) __popcll(um & ((1ULL << lane) - 1ULL))] = qoff;
        }
    }
    if (!FROM_LIST) break;
    __builtin_amdgcn_wave_barrier();
  }
}

#ifdef ME_AB  // measurement build only (make -C profiles/ab): the matrix-pipe ranking that was measured and not adopted
#include "me_nn_kernels_ab.inc"
#endif

// ------------------------------------------------------------------------------------------------------------
// General path: nearest-first walk of the sparse octree (see the header).  `list` == nullptr: every query of
// [q_begin, q_end); otherwise the queries q_begin + list[t], t < *list_count, starting from the bound the grid pass
// left behind.
// ------------------------------------------------------------------------------------------------------------
// Eight lanes ("octet") cooperate on one query: lane c bounds child c of the current node (one 32-byte record each,
// one coalesced 256-byte burst per octet), a 3-step butterfly picks the nearest admissible child, and a leaf cell is
// scanned eight points at a time.  The queries that reach this kernel are the rare far ones (0.1 % of the bench scene,
// but each sweeps hundreds of nodes): with one lane per query the whole launch waited on a handful of serial walks.
__device__ __forceinline__ double octet_min(double v) {  // (DPP exchanges, me_internal.hpp)
    v = fmin(v, octet_partner_d<0>(v));
    v = fmin(v, octet_partner_d<1>(v));
    v = fmin(v, octet_partner_d<2>(v));
    return v;
}

constexpr int kNn1Block = 64;  // 8 octets per block (one wavefront): 8 x (levels + 1) x (8 floats + 9 ints) x 4 B of walk cache, 6 KB at 10 levels
// (96 VGPRs, five wavefronts per SIMD: at 80 / 72 / 64 registers the compiler spills 19 / 27 / 41 dwords and the launch is slower —
// 0.86 / 0.93 / 1.00 / 1.40 ms on the bench pair: the kernel is bound by instruction issue, not by the walks in flight)
constexpr int kNn1Waves = 5;
// COV / DBG (round 5): the covered-band bound of the cross-rank step and the walk counters of the instrumented runs are compile-time
// variants — their live values (the band, seven counters) cost the plain walk, the one every 1-NN pass of a product run launches, 24
// bytes of scratch per lane at 96 VGPRs through round 4.
template <bool COV, bool DBG>
__global__ void __launch_bounds__(kNn1Block) __attribute__((amdgpu_waves_per_eu(kNn1Waves, 8)))
k_nn1(const SPoint *__restrict__ qsp, long long q_begin, long long q_end, const SPoint *__restrict__ rsp, long long nr,
      OctView oct, double *__restrict__ d2_out, int *__restrict__ idx_out, const unsigned int *__restrict__ list,
      const unsigned int *__restrict__ list_count, int use_bound, unsigned long long *__restrict__ dbg,
      unsigned int *__restrict__ far_list, unsigned int *__restrict__ far_count, int far_cap, unsigned int *__restrict__ work,
      const double *__restrict__ cov, int cov_axis) {
    // cov: per query [lo, hi) along cov_axis that its owner has searched already (box_lower_bound_cov); nullptr = none
    // work: a zeroed counter.  A wavefront's first group of eight queries is its block index, every further one is drawn from
    // the counter (round 4; one resident wavefront per slot of the chip).  On the bench pair this changed nothing measurable
    // (0.92 -> 0.96 ms: the launch is bound by instruction issue, profiles/EXPERIMENTS.md) — it is kept for lists whose walks
    // are of very unequal length, where a static deal makes the launch as long as the unluckiest pair of groups.
    // far_list: a walk that has taken `far_cap` steps is ABANDONED here — its best so far is stored as usual, a valid upper
    // bound — and its query appended to far_list for k_nn_far (a whole wavefront per query, big leaves scanned 64 points
    // abreast): the longest chain of dependent steps in this kernel is far_cap, not the walk of the farthest outlier.
    __shared__ long long s_off[kMaxLevels];
    // walk cache, sized by the octree's REAL depth at launch (dynamic LDS: nn1_cache_bytes): per octet and level the children's
    // lower bounds [8] and their [begin, end) on the level below [9].  With the table's worst-case depth (18 levels, 20 KB per
    // block) a CU held 7 blocks; a 50 M-point cloud has ~10 levels.
    extern __shared__ unsigned int s_dyn[];
    if (threadIdx.x < kMaxLevels) s_off[threadIdx.x] = oct.off[threadIdx.x];
    __syncthreads();
    const int L = oct.n_levels - 1;  // root level
    const int lv = oct.n_levels + 1;  // cache lines per octet
    const ONode *__restrict__ nodes = oct.nodes;
    const long long n_items = list ? (long long) *list_count : (q_end - q_begin);
    const int sub = threadIdx.x & 7;
    const unsigned long long t_wave0 = DBG ? wall_clock64() : 0ULL;  // (100 MHz)
    unsigned long long w_open = 0, w_scan = 0, w_pts = 0;  // (DBG only)
    unsigned int w_max = 0;
    static_assert(kNn1Block == 64, "one wavefront per block: the group counter is drawn by lane 0 of the block");
    for (long long g = blockIdx.x;;) {
        const long long t = g * 8 + (threadIdx.x >> 3);
        const bool alive = t < n_items;
        if (!__ballot(alive)) break;  // wave-uniform exit
        {  // the next group, drawn now so that the atomic's round trip hides under this group's walk
            unsigned int nx = 0;
            if (threadIdx.x == 0) nx = atomicAdd(work, 1u);
            g = (long long) gridDim.x + (long long) (unsigned int) __builtin_amdgcn_readfirstlane((int) nx);
        }
        const long long i = q_begin + (alive ? (list ? (long long) list[t] : t) : 0);
        const SPoint q = qsp[i];
        const double qx = q.x, qy = q.y, qz = q.z;
        double clo = INFINITY, chi = -INFINITY;  // (empty interval: nothing covered)
        if (COV && alive) {
            clo = cov[2 * (i - q_begin)];
            chi = cov[2 * (i - q_begin) + 1];
        }
        unsigned int n_open = 0, n_scan = 0;  // nodes opened / point runs scanned by this octet: n_open counts BOTH unless DBG
        unsigned long long n_pts = 0;         // (DBG only)
        double best = INFINITY;
        long long best_i = 0x7fffffffffffffffLL;
        if (list && alive) {
            best = d2_out[i];
            const int bi = idx_out[i];
            if (bi >= 0) best_i = bi;
        } else if (use_bound && alive) {
            best = d2_out[i];  // caller's upper bound (me_nn_points_bounded): only closer points are of interest
        }
        // Per-LANE running best over every scan of this query (round 4, as k_nn_far does): the octet-wide (distance, index)
        // butterfly is paid once per query, not once per scanned run; between scans the walk only needs the pruning bound, an
        // upper bound of the octet's minimum: the lanes' bests rounded UP to FP32 and min-reduced with three DPP moves.
        double lb = best;
        long long li = best_i;
        double bound = best;  // pruning bound: min(best found, tightest box upper bound seen)
        // scan of one run of sorted points [jb, je) by the octets flagged `go`, sixteen points per trip (two independent loads
        // per lane in flight; clamped addresses keep them unconditional)
        auto scan_points = [&](bool go, long long jb, long long je) {
            if (!go) jb = je = 0;
            if (DBG) {
                n_scan += go ? 1u : 0u;
                n_pts += (unsigned long long) (je - jb);
            } else {
                n_open += go ? 1u : 0u;  // (one step counter for the hand-over to k_nn_far)
            }
            for (long long j = jb + sub; __ballot(j < je); j += 16) {
                const long long j1 = j + 8;
                const long long last = je > 0 ? je - 1 : 0;
                const SPoint p0 = rsp[j < je ? j : last];
                const SPoint p1 = rsp[j1 < je ? j1 : last];
                const double d0 = dist2_exact(qx, qy, qz, p0.x, p0.y, p0.z);
                const double d1 = dist2_exact(qx, qy, qz, p1.x, p1.y, p1.z);
                if (j < je && (d0 < lb || (d0 == lb && p0.idx < li))) {  // ties -> smallest reference index (as the oracle)
                    lb = d0;
                    li = p0.idx;
                }
                if (j1 < je && (d1 < lb || (d1 == lb && p1.idx < li))) {
                    lb = d1;
                    li = p1.idx;
                }
            }
            // (squared distances are >= 0: their FP32 images order like their bit patterns)
            int m = __float_as_int(__double2float_ru(lb));
            m = min(m, octet_partner_i<0>(m));
            m = min(m, octet_partner_i<1>(m));
            m = min(m, octet_partner_i<2>(m));
            if (go) bound = fmin(bound, (double) __int_as_float(m));
        };
        bool far_flag = false;
        if (L == 0) {
            scan_points(alive, 0, nr);  // the whole cloud is one cell
        } else {
            // Walk state: level l = the level of the node whose CHILDREN (level l-1) are being considered.  Lane `sub` owns
            // child `sub`: after the one burst that fetches the <= 8 child records (+ the `begin` of the record after
            // each, i.e. the child's end) it keeps, per level, the child's lower bound (a float rounded DOWN: still a
            // lower bound) and its [begin, end) on the level below in a block-local LDS cache.  Returning to a parent therefore
            // costs no memory round trip at all, and a descent costs one.
            // Round 4: the octets of a wavefront that DESCEND in an iteration issue their child-record loads BEFORE the other
            // octets scan their leaf cells, so the iteration waits for one round trip, not two; the block is one wavefront with
            // 6 KB of cache (20 wavefronts per CU in flight instead of 16).
            float *c_lb = reinterpret_cast<float *>(s_dyn) + (threadIdx.x >> 3) * lv * 8;  // [level][8]
            unsigned int *c_beg = s_dyn + (kNn1Block / 8) * lv * 8 + (threadIdx.x >> 3) * lv * 9;  // [level][9]: child begins + the end of the last one
            unsigned long long taken_lo = 0, taken_hi = 0;  // "children already entered" per level: levels 1..8 / 9..16
            bool walking = alive;
            int l = L;
            // the children [cb, ce) of a node, for level lev's cache line; lane `sub` bounds child `sub`.  Two halves: the loads
            // (issued early) and the bounds (after whatever else the iteration has to wait for)
            struct Fetched {
                float4 a, bb;
                unsigned int nxt;
            };
            auto open_load = [&](bool go, int lev, long long cb) {
                if (!go) {  // (an idle octet's cb may be a POINT index of the leaf it is about to scan)
                    cb = 0;
                    lev = 1;
                }
                const long long at = s_off[lev - 1] + cb + sub;
                const float4 *__restrict__ g = reinterpret_cast<const float4 *>(nodes + at);
                // (the buffer has 8 records of slack: short groups are masked, not skipped)
                Fetched f;
                f.a = g[0];
                f.bb = g[1];
                f.nxt = nodes[at + 1].begin;
                return f;
            };
            auto open_finish = [&](bool go, int lev, long long cb, long long ce, const Fetched &ft) {
                if (!go) {
                    cb = ce = 0;
                    lev = 1;
                }
                const int cnt = (int) (ce - cb);
                n_open += go ? 1u : 0u;
                const float f[6] = {ft.a.x, ft.a.y, ft.a.z, ft.a.w, ft.bb.x, ft.bb.y};
                const bool mine = go && sub < cnt;
                // every child box also yields an UPPER bound on the answer (some point lies inside it, no farther than
                // its farthest corner): keeps the depth-first walk from sweeping a wide region on a loose `best`
                const double ub = octet_min(mine ? box_upper_bound(f, qx, qy, qz) : INFINITY);
                const double lbd = COV ? box_lower_bound_cov(f, qx, qy, qz, cov_axis, clo, chi) : box_lower_bound(f, qx, qy, qz);
                if (go) {
                    bound = fmin(bound, ub);
                    c_lb[lev * 8 + sub] = mine ? __double2float_rd(lbd) : INFINITY;
                    c_beg[lev * 9 + sub] = __float_as_uint(ft.bb.z);  // ONode::begin
                    if (sub == 7 || sub == cnt - 1) c_beg[lev * 9 + sub + 1] = ft.nxt;
                }
            };
            {
                const ONode *__restrict__ root = nodes + s_off[L];
                const long long rb = root[0].begin, re = root[1].begin;
                const Fetched ft = open_load(walking, L, rb);
                open_finish(walking, L, rb, re, ft);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bool far = false;
            while (__ballot(walking)) {
                if (far_list && walking && n_open + n_scan >= (unsigned int) far_cap) {
                    far = true;
                    walking = false;
                }
                const unsigned int tk = (l <= 8) ? (unsigned int) (taken_lo >> (8 * (l - 1))) & 0xffu
                                                 : (unsigned int) (taken_hi >> (8 * (l - 9))) & 0xffu;
                const float lbf = walking ? c_lb[l * 8 + sub] : INFINITY;
                const bool ok = walking && !((tk >> sub) & 1u) && (double) lbf <= bound;  // <=: ties may hold a smaller index
                // nearest admissible child: (lower bound, child) packed so that one integer minimum picks it (bounds are >= 0:
                // their bit patterns order like the values)
                unsigned long long key = ok ? (((unsigned long long) __float_as_uint(lbf)) << 3) | (unsigned int) sub : ~0ULL;
                auto kmin = [&](unsigned long long o) { key = o < key ? o : key; };
                kmin((unsigned long long) octet_partner_ll<0>((long long) key));
                kmin((unsigned long long) octet_partner_ll<1>((long long) key));
                kmin((unsigned long long) octet_partner_ll<2>((long long) key));
                const int kc = key == ~0ULL ? 8 : (int) (key & 7u);
                bool go_leaf = false, go_down = false;
                long long cb = 0, ce = 0;
                if (walking) {
                    if (kc >= 8) {  // nothing left under this node: back to the parent's cache line
                        if (l == L) walking = false;
                        else ++l;
                    } else {
                        if (l <= 8) taken_lo |= 1ULL << (8 * (l - 1) + kc);
                        else taken_hi |= 1ULL << (8 * (l - 9) + kc);
                        cb = c_beg[l * 9 + kc];
                        ce = c_beg[l * 9 + kc + 1];
                        if (l == 1) {
                            go_leaf = true;  // [cb, ce) are the points of a leaf cell
                        } else {
                            go_down = true;  // [cb, ce) are the chosen child's children, on level l - 2
                            --l;
                            if (l <= 8) taken_lo &= ~(0xffULL << (8 * (l - 1)));  // fresh node on the level below
                            else taken_hi &= ~(0xffULL << (8 * (l - 9)));
                        }
                    }
                }
                const bool any_down = __ballot(go_down) != 0ULL;
                Fetched ft;
                ft.a = ft.bb = make_float4(0.f, 0.f, 0.f, 0.f);
                ft.nxt = 0u;
                if (any_down) ft = open_load(go_down, go_down ? l : 1, cb);
                if (__ballot(go_leaf)) scan_points(go_leaf, cb, ce);
                if (any_down) {
                    open_finish(go_down, go_down ? l : 1, cb, ce, ft);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            far_flag = far;
        }
        {  // the query's answer: exact minimum over the octet's lanes, ties -> smallest reference index
            auto stage = [&](double od, long long oi) {
                if (od < lb || (od == lb && oi < li)) {
                    lb = od;
                    li = oi;
                }
            };
            stage(octet_partner_d<0>(lb), octet_partner_ll<0>(li));
            stage(octet_partner_d<1>(lb), octet_partner_ll<1>(li));
            stage(octet_partner_d<2>(lb), octet_partner_ll<2>(li));
            best = lb;
            best_i = li;
        }
        {  // abandoned walks go to the far list (wave-aggregated append)
            const unsigned long long fm = __ballot(far_flag && sub == 0);
            if (fm) {
                unsigned int base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(far_count, (unsigned int) __popcll(fm));
                base = (unsigned int) __builtin_amdgcn_readfirstlane((int) base);
                if (far_flag && sub == 0)
                    far_list[base + (unsigned int) __popcll(fm & ((1ULL << (threadIdx.x & 63)) - 1ULL))] = (unsigned int) (i - q_begin);
            }
        }
        if (alive && sub == 0) {
            d2_out[i] = best;
            idx_out[i] = (int) best_i;
            if (DBG) {
                w_open += n_open;  // (profiling counters: summed per wavefront, see the end of the kernel)
                w_scan += n_scan;
                w_pts += n_pts;
                w_max = max(w_max, n_open + n_scan);
            }
        }
    }  // grid-stride loop over octets
    if (DBG && dbg) {
        // me_timer_get "nn1_opened" / "nn1_scans" / "nn1_points" / "nn1_max_opened" (only while timers are on).  ONE atomic per
        // wavefront and counter: an atomic per query — 470 k of them on four addresses, serialised in one L2 channel — cost the
        // instrumented launch 0.8 ms of its 1.65 (round 4; the untimed steps never paid it)
        for (int m = 8; m < 64; m <<= 1) {
            w_open += __shfl_xor(w_open, m, 64);
            w_scan += __shfl_xor(w_scan, m, 64);
            w_pts += __shfl_xor(w_pts, m, 64);
            w_max = max(w_max, (unsigned int) __shfl_xor((int) w_max, m, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            const unsigned long long dt = wall_clock64() - t_wave0;  // "nn1_wave_max_10ns" / "nn1_wave_sum_10ns" / "nn1_waves"
            if (w_open | w_scan) {
                atomicAdd(&dbg[0], w_open);
                atomicAdd(&dbg[1], w_scan);
                atomicAdd(&dbg[2], w_pts);
                atomicMax(&dbg[3], (unsigned long long) w_max);
                atomicMax(&dbg[8], dt);
                atomicAdd(&dbg[9], dt);
                atomicAdd(&dbg[11], 1ULL);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_nn_far — the walks k_nn1 abandoned (far_list), ONE WAVEFRONT PER QUERY.
// A query far from the reference surface (an outlier metres above the ground) must look into every leaf cell that its
// current best distance reaches: a disc of hundreds of 6-point cells, one dependent step each in k_nn1 (734 steps for the
// worst query of the bench pair, ~3 us per step: that ONE walk was the kernel's duration).  Here the walk stops descending
// at a node that holds at most `far_leaf` points — a contiguous run of the sorted array, `pbegin` — and scans it with all 64
// lanes, four independent loads per lane in flight: a handful of dependent steps and a few streaming scans per query.
// Same pruning rule as k_nn1 (a child is entered while its lower bound is <= the bound; ties -> smallest reference
// index), the walk restarts from the root with the abandoned walk's best as its bound: the result is the exact minimum.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_nn_far(const SPoint *__restrict__ qsp, long long q_begin, const SPoint *__restrict__ rsp, OctView oct,
         double *__restrict__ d2_out, int *__restrict__ idx_out, const unsigned int *__restrict__ far_list,
         const unsigned int *__restrict__ far_count, int far_leaf, unsigned long long *__restrict__ dbg,
         const double *__restrict__ cov, int cov_axis) {
    __shared__ long long s_off[kMaxLevels];
    __shared__ float s_lb[4][kMaxLevels + 1][8];
    __shared__ unsigned int s_beg[4][kMaxLevels + 1][9], s_pb[4][kMaxLevels + 1][9];
    if (*far_count == 0) return;  // the common case (and every slab rank's): nothing was handed over
    if (threadIdx.x < kMaxLevels) s_off[threadIdx.x] = oct.off[threadIdx.x];
    __syncthreads();
    const int L = oct.n_levels - 1;
    if (L == 0) return;  // (a one-cell cloud never abandons a walk)
    const ONode *__restrict__ nodes = oct.nodes;
    const unsigned int *__restrict__ pbegin = oct.pbegin;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *c_lb = s_lb[wv][0];
    unsigned int *c_beg = s_beg[wv][0], *c_pb = s_pb[wv][0];
    const unsigned int n_far = *far_count;
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&dbg[4], (unsigned long long) n_far);  // "nn1_far"
    for (unsigned int t = blockIdx.x * 4u + (unsigned int) wv; t < n_far; t += gridDim.x * 4u) {
        const long long i = q_begin + (long long) far_list[t];
        const SPoint q = qsp[i];
        const double qx = q.x, qy = q.y, qz = q.z;
        const double clo = cov ? cov[2 * (i - q_begin)] : INFINITY, chi = cov ? cov[2 * (i - q_begin) + 1] : -INFINITY;
        double best = d2_out[i];
        long long best_i = idx_out[i] >= 0 ? (long long) idx_out[i] : 0x7fffffffffffffffLL;
        double bound = best;
        // per-lane running best over every scan of this query (round 3): the wave-wide (distance, index) butterfly — 24
        // ds_bpermute round trips — is paid ONCE per query, not once per scanned node; between scans only the pruning bound is
        // needed, an upper bound of the wave's minimum: the lanes' bests rounded UP to FP32 and min-reduced with DPP moves
        double lb = best;
        long long li = best_i;
        unsigned long long taken_lo = 0, taken_hi = 0;  // wave-uniform
        int l = L;
        unsigned int st_open = 0, st_scan = 0;  // (profiling counters, `dbg`)
        unsigned long long st_pts = 0;
        // children [cb, ce) of a node -> level lev's cache line (lanes 0..7 bound one child each)
        auto open_node = [&](int lev, unsigned int cb, unsigned int ce) {
            ++st_open;
            const int cnt = (int) (ce - cb);
            double ub = INFINITY;
            if (lane < 8) {
                const long long at = s_off[lev - 1] + (long long) cb + lane;  // (8 records of slack behind the last level)
                const float4 *__restrict__ g = reinterpret_cast<const float4 *>(nodes + at);
                const float4 a = g[0], bb = g[1];
                const float f[6] = {a.x, a.y, a.z, a.w, bb.x, bb.y};
                const bool mine = lane < cnt;
                if (mine) ub = box_upper_bound(f, qx, qy, qz);
                c_lb[lev * 8 + lane] = mine ? __double2float_rd(cov ? box_lower_bound_cov(f, qx, qy, qz, cov_axis, clo, chi) : box_lower_bound(f, qx, qy, qz)) : INFINITY;
                c_beg[lev * 9 + lane] = __float_as_uint(bb.z);
                c_pb[lev * 9 + lane] = pbegin[at];
                if (lane == 7 || lane == cnt - 1) {
                    c_beg[lev * 9 + lane + 1] = nodes[at + 1].begin;
                    c_pb[lev * 9 + lane + 1] = pbegin[at + 1];
                }
            }
            ub = octet_min(ub);  // (lanes 8..63 hold +inf: their octets reduce to +inf)
            bound = fmin(bound, uniform_f64(ub));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        {
            const ONode *__restrict__ root = nodes + s_off[L];
            open_node(L, root[0].begin, root[1].begin);
        }
        for (;;) {
            const unsigned int tk = (l <= 8) ? (unsigned int) (taken_lo >> (8 * (l - 1))) & 0xffu
                                             : (unsigned int) (taken_hi >> (8 * (l - 9))) & 0xffu;
            const double lbd = lane < 8 ? (double) c_lb[l * 8 + lane] : INFINITY;
            const bool ok = lane < 8 && !((tk >> lane) & 1u) && lbd <= bound;  // <=: ties may hold a smaller index
            double kd = ok ? lbd : INFINITY;
            int kc = ok ? lane : 8;
            auto pick = [&](double od, int oc) {
                if (od < kd || (od == kd && oc < kc)) {
                    kd = od;
                    kc = oc;
                }
            };
            pick(octet_partner_d<0>(kd), octet_partner_i<0>(kc));
            pick(octet_partner_d<1>(kd), octet_partner_i<1>(kc));
            pick(octet_partner_d<2>(kd), octet_partner_i<2>(kc));
            kc = __builtin_amdgcn_readfirstlane(kc);
            if (kc >= 8) {  // nothing left under this node
                if (l == L) break;
                ++l;
                continue;
            }
            if (l <= 8) taken_lo |= 1ULL << (8 * (l - 1) + kc);
            else taken_hi |= 1ULL << (8 * (l - 9) + kc);
            const unsigned int cb = c_beg[l * 9 + kc], ce = c_beg[l * 9 + kc + 1];
            const unsigned int pb = c_pb[l * 9 + kc], pe = c_pb[l * 9 + kc + 1];
            if (l == 1 || pe - pb <= (unsigned int) far_leaf) {
                ++st_scan;
                st_pts += pe - pb;
                // scan the node's points, 4 x 64 at a time (clamped addresses keep the four loads unconditional)
                for (unsigned int j0 = pb; j0 < pe; j0 += 256u) {
                    SPoint p[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned int j = j0 + 64u * u + (unsigned int) lane;
                        p[u] = rsp[j < pe ? j : pe - 1u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned int j = j0 + 64u * u + (unsigned int) lane;
                        const double d = dist2_exact(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                        if (j < pe && (d < lb || (d == lb && p[u].idx < li))) {
                            lb = d;
                            li = p[u].idx;
                        }
                    }
                }
                // (squared distances are >= 0: their FP32 images order like their bit patterns)
                const float wmin = __int_as_float(wave_min_i(__float_as_int(__double2float_ru(lb))));
                bound = fmin(bound, (double) wmin);
            } else {  // [cb, ce) are the chosen child's children, on level l - 2
                --l;
                if (l <= 8) taken_lo &= ~(0xffULL << (8 * (l - 1)));
                else taken_hi &= ~(0xffULL << (8 * (l - 9)));
                open_node(l, cb, ce);
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {  // the query's answer: exact minimum, ties -> smallest reference index
            const double od = __shfl_xor(lb, m, 64);
            const long long oi = __shfl_xor(li, m, 64);
            if (od < lb || (od == lb && oi < li)) {
                lb = od;
                li = oi;
            }
        }
        best = lb;
        best_i = li;
        if (lane == 0) {
            d2_out[i] = best;
            idx_out[i] = (int) best_i;
            if (dbg) {  // (me_timer_get "nn1_far_opened" / "nn1_far_points" / "nn1_far_max"; only while timers are on)
                atomicAdd(&dbg[5], (unsigned long long) st_open);
                atomicAdd(&dbg[6], st_pts);
                atomicMax(&dbg[7], (unsigned long long) (st_open + st_scan));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- un-permute results to the caller's (original) query order ----
__global__ void k_nn_unpermute(const SPoint *__restrict__ qsp, long long q_begin, long long q_end,
                               const double *__restrict__ d2s, const int *__restrict__ idxs, double *__restrict__ d2o,
                               int *__restrict__ idxo) {
    const long long i = q_begin + (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q_end) return;
    const long long o = qsp[i].idx;
    if (d2o) d2o[o] = d2s[i];
    if (idxo) idxo[o] = idxs[i];
}

// ---- statistics ----
struct StatParams {
    double gate;      // threshold on d2 (already squared if the mode says so); < 0 = no gate
    int gate_strict;  // 1: d2 < gate, 0: d2 <= gate
    double t2max[5];  // largest d2 whose correctly rounded sqrt is <= trunc[k]
};

constexpr int kStatD = 11;  // sum_d[5], sum_d2[5], sum_sqrt_all
constexpr int kStatI = 6;   // n_corr, n_inl[5]

__device__ __forceinline__ bool gate_pass(const StatParams &sp, double d2) {
    if (sp.gate < 0) return true;
    return sp.gate_strict ? (d2 < sp.gate) : (d2 <= sp.gate);
}

__global__ void __launch_bounds__(256)
k_nn_partial(const double *__restrict__ d2s, long long q_begin, long long q_end, StatParams sp,
             double *__restrict__ pd, long long *__restrict__ pi) {
    double sd[kStatD];
    long long si[kStatI];
#pragma unroll
    for (int k = 0; k < kStatD; ++k) sd[k] = 0;
#pragma unroll
    for (int k = 0; k < kStatI; ++k) si[k] = 0;
    auto take = [&](double d2) {
        if (d2 < 0.0) return;       // slab mode: halo point, not a query of this rank
        const double d = sqrt(d2);  // (map_pt - gt_pt).norm(), map_eval.cpp:1095
        sd[10] += d;                // computeChamferDistance, ungated (map_eval.cpp:1416)
        if (gate_pass(sp, d2)) {
            si[0] += 1;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (d2 <= sp.t2max[k]) {  // <=> norm_dis <= trunc_dist_[k] (map_eval.cpp:1099-1123)
                    sd[k] += d;
                    sd[5 + k] += d2;
                    si[1 + k] += 1;
                }
        }
    };
    // (round 6: four loads in flight per thread — a thread walks ~190 strided elements of a 50 M-point pass, one dependent L2 / HBM
    // round trip each before; the elements are taken in the same order, so every sum is bit for bit what it was)
    const long long S = (long long) gridDim.x * blockDim.x;
    long long i = q_begin + (long long) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * S < q_end; i += 4 * S) {
        const double a0 = d2s[i], a1 = d2s[i + S], a2 = d2s[i + 2 * S], a3 = d2s[i + 3 * S];
        take(a0);
        take(a1);
        take(a2);
        take(a3);
    }
    for (; i < q_end; i += S) take(d2s[i]);
    __shared__ double smd[4];
    __shared__ long long smi[4];
#pragma unroll
    for (int k = 0; k < kStatD; ++k) {
        const double r = block_sum_256(sd[k], smd);
        if (threadIdx.x == 0) pd[(long long) blockIdx.x * kStatD + k] = r;
    }
#pragma unroll
    for (int k = 0; k < kStatI; ++k) {
        const long long r = block_sum_256_ll(si[k], smi);
        if (threadIdx.x == 0) pi[(long long) blockIdx.x * kStatI + k] = r;
    }
}

struct Mean5 {
    double m[5];
};

__global__ void __launch_bounds__(256)
k_nn_sigma(const double *__restrict__ d2s, long long q_begin, long long q_end, StatParams sp, Mean5 mean,
           double *__restrict__ pd) {
    double s[5] = {0, 0, 0, 0, 0};
    auto take = [&](double d2) {
        if (d2 >= 0.0 && gate_pass(sp, d2)) {
            const double d = sqrt(d2);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double e = d - mean.m[k];  // every correspondence, not only the inliers (map_eval.cpp:1133-1136)
                s[k] += e * e;
            }
        }
    };
    const long long S = (long long) gridDim.x * blockDim.x;  // (four loads in flight, same order of summation: see k_nn_partial)
    long long i = q_begin + (long long) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * S < q_end; i += 4 * S) {
        const double a0 = d2s[i], a1 = d2s[i + S], a2 = d2s[i + 2 * S], a3 = d2s[i + 3 * S];
        take(a0);
        take(a1);
        take(a2);
        take(a3);
    }
    for (; i < q_end; i += S) take(d2s[i]);
    __shared__ double smd[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double r = block_sum_256(s[k], smd);
        if (threadIdx.x == 0) pd[(long long) blockIdx.x * 5 + k] = r;
    }
}

// deterministic final reduction: one 256-thread block per component, fixed strided partial sums + fixed tree
// (a single lane per component used to walk the <= 1024 block partials serially: 0.15 ms of pure latency per call)
__global__ void __launch_bounds__(256)
k_final_sum_d(const double *__restrict__ part, int nblocks, int ncomp, double *__restrict__ out) {
    const int k = blockIdx.x;
    double s = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(long long) b * ncomp + k];
    __shared__ double sm[4];
    const double r = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[k] = r;
}
__global__ void __launch_bounds__(256)
k_final_sum_i(const long long *__restrict__ part, int nblocks, int ncomp, long long *__restrict__ out) {
    const int k = blockIdx.x;
    long long s = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(long long) b * ncomp + k];
    __shared__ long long sm[4];
    const long long r = block_sum_256_ll(s, sm);
    if (threadIdx.x == 0) out[k] = r;
}

static StatParams make_params(double gate, int gate_mode, const double trunc[5]) {
    StatParams sp;
    if (gate < 0) {
        sp.gate = -1.0;
        sp.gate_strict = 0;
    } else if (gate_mode == ME_GATE_LT_SQUARED) {
        sp.gate = gate * gate;
        sp.gate_strict = 1;
    } else {
        sp.gate = gate;
        sp.gate_strict = 0;
    }
    for (int k = 0; k < 5; ++k) {
        // largest double x with sqrt_rn(x) <= t, so the device compares d2 against it and the inlier count does
        // not depend on the device's sqrt rounding
        const double t = trunc ? trunc[k] : 0.0;
        if (!(t >= 0)) {
            sp.t2max[k] = -1.0;
            continue;
        }
        double x = t * t;
        while (std::sqrt(std::nextafter(x, INFINITY)) <= t) x = std::nextafter(x, INFINITY);
        while (x > 0 && std::sqrt(x) > t) x = std::nextafter(x, -INFINITY);
        sp.t2max[k] = x;
    }
    return sp;
}

// ---- slab mode helpers -------------------------------------------------------------------------------------
// the reference slab of this rank is empty: every owned query is unresolved with an infinite bound
__global__ void k_nn_fill_empty(const SPoint *__restrict__ qsp, long long n, SlabView slab, double *__restrict__ d2_out,
                                int *__restrict__ idx_out) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SPoint q = qsp[i];
    d2_out[i] = slab_owned(slab, q.x, q.y, q.z) ? INFINITY : -1.0;
    idx_out[i] = -1;
}

// owned queries whose best distance is not below their distance to the nearest outer face of (slab + halo): a closer
// reference point may live on another rank
__global__ void k_nn_collect_unresolved(const SPoint *__restrict__ qsp, long long n, SlabView slab,
                                        const double *__restrict__ d2, unsigned int *__restrict__ list,
                                        unsigned int *__restrict__ count) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    bool un = false;
    if (i < n) {
        const double d = d2[i];
        if (d >= 0.0) {  // owned
            const SPoint q = qsp[i];
            const double f = slab_face_distance(slab, q.x, q.y, q.z);
            un = !(d < f * f);
        }
    }
    const unsigned long long um = __ballot(un);
    if (um) {
        const int lane = threadIdx.x & 63;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned int) __popcll(um));
        base = (unsigned int) readlane_i((int) base, 0);
        if (un) list[base + (unsigned int) __popcll(um & ((1ULL << lane) - 1ULL))] = (unsigned int) i;
    }
}

__global__ void k_nn_export_xyz(const SPoint *__restrict__ qsp, const unsigned int *__restrict__ list, long long m,
                                double *__restrict__ xyz, const double *__restrict__ d2s, double *__restrict__ d2) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const SPoint q = qsp[list[t]];
    if (d2) d2[t] = d2s[list[t]];
    xyz[3 * t] = q.x;
    xyz[3 * t + 1] = q.y;
    xyz[3 * t + 2] = q.z;
}

__global__ void k_points_to_sp(const double *__restrict__ xyz, long long m, SPoint *__restrict__ sp) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    SPoint p;
    p.x = xyz[3 * t];
    p.y = xyz[3 * t + 1];
    p.z = xyz[3 * t + 2];
    p.idx = t;
    sp[t] = p;
}

__global__ void k_fill_f64(double *p, long long n, double v) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = v;
}

__global__ void k_nn_patch(const unsigned int *__restrict__ list, long long m, const double *__restrict__ d2_new,
                           double *__restrict__ d2) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const unsigned int i = list[t];
    d2[i] = fmin(d2[i], d2_new[t]);
}

// steps after which k_nn1 hands a walk over to k_nn_far (0 = never: the first version's behaviour)
static constexpr int nn1_far_cap() { return ME_TUNE_NN1_FAR_CAP <= 0 ? 0x7fffffff : ME_TUNE_NN1_FAR_CAP; }
// points a node may hold for k_nn_far to scan it whole instead of descending further
static constexpr int nn_far_leaf() { return ME_TUNE_NN_FAR_LEAF < 1 ? 1 : ME_TUNE_NN_FAR_LEAF; }
constexpr unsigned int kFarGrid = 1024;  // blocks of four wavefronts striding over the far list (empty list: immediate return)
static size_t nn1_cache_bytes(const OctView &oct) { return (size_t) (kNn1Block / 8) * (size_t) (oct.n_levels + 1) * (8 + 9) * 4; }

int nn_search(me_ctx *ctx, int qslot, int rslot) {
    if (qslot < 0 || qslot > 1 || rslot < 0 || rslot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    Cloud &q = ctx->cloud[qslot];
    Cloud &r = ctx->cloud[rslot];
    const bool slab = q.slab.axis >= 0;
    if (!q.uploaded || !r.uploaded || (!slab && (!q.index_valid || !r.index_valid)))
        return ctx->fail(ME_ERR_STATE, "me_nn1: upload both clouds first");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    ME_TRY(cloud_finish_octree(ctx, rslot));  // (an index built with ctx->defer_octree; me_run_suite_from has done this already)
    q.nn_ref_slot = rslot;
    ME_TRACE_POINT(ctx, "nn_search: enter");
    q.n_unres = 0;
    if (q.n == 0) return ME_OK;  // empty slab: nothing to query
    ME_CHECK(ctx, q.nn_d2.ensure((size_t) q.n * 8));
    ME_CHECK(ctx, q.nn_idx.ensure((size_t) q.n * 4));
    ME_CHECK(ctx, ctx->red.ensure(64));
    unsigned int *d_cnt = ctx->red.as<unsigned int>();
    long long b, e;
    ctx->shard_range(q.n, b, e);
    if (r.n == 0) {
        hipLaunchKernelGGL(k_nn_fill_empty, dim3((unsigned int) ((q.n + 255) / 256)), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(),
                           q.n, q.slab, q.nn_d2.as<double>(), q.nn_idx.as<int>());
    } else if (e > b) {
        const unsigned int nb = (unsigned int) (((e - b + 255) / 256 + 7) / 8 * 8);  // multiple of 8 (XCD chunking)
        ME_CHECK(ctx, q.nn_list.ensure((size_t) (e - b) * 4 + 64));
        ME_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 32, ctx->stream));  // [0] unresolved-list length, [1] far-list length, [4] k_nn1's group counter, [2] first-pass list
        FrameView fr{r.origin[0], r.origin[1], r.origin[2], r.fine_h};
        // Cascade: fine grid -> (what it leaves) every coarser level up to the radius grid -> (what that leaves) the octree.
        const GridView *levels[Cloud::kMaxMid + 1];
        int n_levels = 0;
        if (r.grid.cell_start && r.grid.shift > r.nn_grid.shift) {
            for (int k = 0; k < r.n_mid; ++k)
                if (r.mid_grid[k].shift > r.nn_grid.shift && r.mid_grid[k].shift < r.grid.shift) levels[n_levels++] = &r.mid_grid[k];
            levels[n_levels++] = &r.grid;
        }
        const bool two_pass = n_levels > 0;
        if (two_pass) {
            ME_CHECK(ctx, ctx->nn_flags.ensure((size_t) (e - b) + 64));
            ME_CHECK(ctx, ctx->nn_list_a.ensure((size_t) (e - b) * 4 + 64));
            if (n_levels > 1) ME_CHECK(ctx, ctx->nn_list_b.ensure((size_t) (e - b) * 4 + 64));
        }
        unsigned int *list_a = two_pass ? ctx->nn_list_a.as<unsigned int>() : q.nn_list.as<unsigned int>();
        unsigned int *cnt_a = two_pass ? d_cnt + 2 : d_cnt;
        {
            TimerScope ts(ctx, "nn_grid");
#ifdef ME_AB
#include "me_nn_dispatch_ab.inc"
#endif
            {
                hipLaunchKernelGGL((k_nn_grid<false>), dim3(nb), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(), b, e, r.sp.as<SPoint>(),
                                   r.n, r.nn_grid, fr, q.slab, q.nn_d2.as<double>(), q.nn_idx.as<int>(), q.nn_list.as<unsigned int>(), d_cnt,
                                   xcd_chunk_setting(), (const unsigned int *) nullptr, (const unsigned int *) nullptr,
                                   two_pass ? ctx->nn_flags.as<unsigned char>() : (unsigned char *) nullptr);
                // (one pass only: the unresolved queries were appended to the octree's list directly, in any order)
                if (two_pass) ME_TRY(select_flagged_u32(ctx, ctx->nn_flags.as<unsigned char>(), e - b, list_a, cnt_a));
            }
        }
        if (two_pass) {
            // level after level: the list of level k (curve-ordered after the compaction, roughly so afterwards: the appends of a
            // pass keep the order of its input up to the scheduling of its waves) -> level k + 1; the last level appends to the
            // octree's list.  [2], [3] of the counter block are the ping-pong lists' lengths.
            TimerScope ts(ctx, "nn_grid2");
            const unsigned int *in_list = list_a, *in_cnt = cnt_a;
            for (int k = 0; k < n_levels; ++k) {
                const bool last = k + 1 == n_levels;
                unsigned int *out_list = last ? q.nn_list.as<unsigned int>() : ((k & 1) ? ctx->nn_list_a.as<unsigned int>() : ctx->nn_list_b.as<unsigned int>());
                unsigned int *out_cnt = last ? d_cnt : ((k & 1) ? d_cnt + 2 : d_cnt + 3);
                if (!last && k >= 1) ME_CHECK(ctx, hipMemsetAsync(out_cnt, 0, 4, ctx->stream));  // (that list was read two passes ago: reuse it)
                hipLaunchKernelGGL((k_nn_grid<true>), dim3(2048), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(), b, e, r.sp.as<SPoint>(), r.n,
                                   *levels[k], fr, q.slab, q.nn_d2.as<double>(), q.nn_idx.as<int>(), out_list, out_cnt, xcd_chunk_setting(),
                                   in_list, in_cnt, (unsigned char *) nullptr);
                in_list = out_list;
                in_cnt = out_cnt;
            }
        }
        {
            // the list length stays on the device: a fixed grid strides over it (no host round trip)
            // one octet per listed query, at most one resident wavefront per slot of the chip (8 octets per block)
            const unsigned int nbf = (unsigned int) std::min<long long>((e - b + 7) / 8, 256 * 4 * kNn1Waves);
            // (the far list can hold every query of the list: sized like it)
            ME_CHECK(ctx, ctx->nn_far.ensure((size_t) (e - b) * 4 + 64));
            {
                TimerScope ts(ctx, "nn1");
#define ME_LAUNCH_NN1(COV_, DBG_, DBGP)                                                                                                  \
    hipLaunchKernelGGL((k_nn1<COV_, DBG_>), dim3(nbf), dim3(kNn1Block), nn1_cache_bytes(r.oct), ctx->stream, q.sp.as<SPoint>(), b, e,        \
                       r.sp.as<SPoint>(), r.n, r.oct, q.nn_d2.as<double>(), q.nn_idx.as<int>(), q.nn_list.as<unsigned int>(), d_cnt, 0, DBGP, \
                       ctx->nn_far.as<unsigned int>(), d_cnt + 1, nn1_far_cap(), d_cnt + 4, (const double *) nullptr, 0)
                unsigned long long *dbgp = ctx->timers_on ? ctx->nn1_dbg() : nullptr;
                if (dbgp) ME_LAUNCH_NN1(false, true, dbgp);
                else ME_LAUNCH_NN1(false, false, (unsigned long long *) nullptr);
#undef ME_LAUNCH_NN1
            }
            {
                TimerScope ts(ctx, "nn_far");
                hipLaunchKernelGGL(k_nn_far, dim3(kFarGrid), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(), b, r.sp.as<SPoint>(), r.oct,
                                   q.nn_d2.as<double>(), q.nn_idx.as<int>(), ctx->nn_far.as<unsigned int>(), d_cnt + 1, nn_far_leaf(),
                                   ctx->timers_on ? ctx->nn1_dbg() : nullptr, (const double *) nullptr, 0);
            }
        }
        if (ctx->timers_on) {  // fallback share, for the bench report
            unsigned int h = 0;
            ME_TRY(copy_d2h(ctx, &h, d_cnt, 4));
            ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->nn_fallback += h;
            ctx->nn_queries += e - b;
        }
    }
    if (slab) {  // which owned queries might have a closer neighbour on another rank?
        ME_CHECK(ctx, q.nn_unres.ensure((size_t) q.n * 4 + 64));
        ME_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_nn_collect_unresolved, dim3((unsigned int) ((q.n + 255) / 256)), dim3(256), 0, ctx->stream,
                           q.sp.as<SPoint>(), q.n, q.slab, q.nn_d2.as<double>(), q.nn_unres.as<unsigned int>(), d_cnt);
        unsigned int h = 0;
        ME_TRY(copy_d2h(ctx, &h, d_cnt, 4));
        ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        q.n_unres = h;
    }
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

int nn_unresolved(me_ctx *ctx, int qslot, double *xyz_device, double *d2_device, long long capacity, long long *count) {
    if (qslot < 0 || qslot > 1 || !count) return ctx->fail(ME_ERR_ARG, "me_nn_unresolved: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "no NN result for this slot (call me_nn1 first)");
    *count = q.n_unres;
    if (!xyz_device || q.n_unres == 0) return ME_OK;
    if (capacity < q.n_unres) return ctx->fail(ME_ERR_CAPACITY, "me_nn_unresolved: capacity too small");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_nn_export_xyz, dim3((unsigned int) ((q.n_unres + 255) / 256)), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(),
                       q.nn_unres.as<unsigned int>(), q.n_unres, xyz_device, q.nn_d2.as<double>(), d2_device);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int nn_points(me_ctx *ctx, int rslot, const double *xyz_device, long long m, double *d2_device, bool bounded, int cov_axis,
              const double *cov_device) {
    if (rslot < 0 || rslot > 1 || m < 0 || (m > 0 && (!xyz_device || !d2_device))) return ctx->fail(ME_ERR_ARG, "me_nn_points: bad argument");
    Cloud &r = ctx->cloud[rslot];
    if (!r.uploaded) return ctx->fail(ME_ERR_STATE, "me_nn_points: reference cloud not uploaded");
    if (m == 0) return ME_OK;
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    ME_TRY(cloud_finish_octree(ctx, rslot));
    const unsigned int nb = (unsigned int) ((m + 255) / 256);
    if (r.n == 0) {
        if (!bounded) hipLaunchKernelGGL(k_fill_f64, dim3(nb), dim3(256), 0, ctx->stream, d2_device, m, (double) INFINITY);
    } else {
        DevBuf &qs = ctx->tmp[0], &qi = ctx->tmp[1];
        ME_CHECK(ctx, qs.ensure((size_t) m * sizeof(SPoint)));
        ME_CHECK(ctx, qi.ensure((size_t) m * 4));
        hipLaunchKernelGGL(k_points_to_sp, dim3(nb), dim3(256), 0, ctx->stream, xyz_device, m, qs.as<SPoint>());
        // caller-supplied points (the cross-rank step, bulk queries): the same hand-over as me_nn1 — a walk that has taken
        // far_cap steps stores its best so far and goes to k_nn_far (the far list holds indices into qs)
        ME_CHECK(ctx, ctx->nn_far.ensure((size_t) m * 4 + 64));
        ME_CHECK(ctx, ctx->red.ensure(64));
        unsigned int *d_cnt = ctx->red.as<unsigned int>();
        ME_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 32, ctx->stream));  // ([4]: k_nn1's group counter)
        TimerScope ts(ctx, "nn1");
#define ME_LAUNCH_NN1P(COV_, DBG_, DBGP)                                                                                                    \
    hipLaunchKernelGGL((k_nn1<COV_, DBG_>), dim3((unsigned int) std::min<long long>((m + 7) / 8, 256 * 4 * kNn1Waves)), dim3(kNn1Block),        \
                       nn1_cache_bytes(r.oct), ctx->stream, qs.as<SPoint>(), 0LL, m, r.sp.as<SPoint>(), r.n, r.oct, d2_device, qi.as<int>(),    \
                       (const unsigned int *) nullptr, (const unsigned int *) nullptr, bounded ? 1 : 0, DBGP, ctx->nn_far.as<unsigned int>(),   \
                       d_cnt + 1, nn1_far_cap(), d_cnt + 4, cov_device, cov_axis)
        unsigned long long *dbgp = ctx->timers_on ? ctx->nn1_dbg() : nullptr;
        if (cov_device && dbgp) ME_LAUNCH_NN1P(true, true, dbgp);
        else if (cov_device) ME_LAUNCH_NN1P(true, false, (unsigned long long *) nullptr);
        else if (dbgp) ME_LAUNCH_NN1P(false, true, dbgp);
        else ME_LAUNCH_NN1P(false, false, (unsigned long long *) nullptr);
#undef ME_LAUNCH_NN1P
        hipLaunchKernelGGL(k_nn_far, dim3(kFarGrid), dim3(256), 0, ctx->stream, qs.as<SPoint>(), 0LL, r.sp.as<SPoint>(), r.oct, d2_device,
                           qi.as<int>(), ctx->nn_far.as<unsigned int>(), d_cnt + 1, nn_far_leaf(), ctx->timers_on ? ctx->nn1_dbg() : nullptr, cov_device, cov_axis);
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The cross-rank 1-NN step of a slab run in three library calls (round 5; dist.py and the C++ host drove it with ~60 small
// tensor operations and two searches + two patches per direction).  One fixed-capacity MESSAGE per rank:
//   row 0                 [open queries map -> gt, open queries gt -> map, local points of the map, of the ground truth]
//   rows 1 .. cap         the open queries of the map -> ground-truth search: x, y, z, bound (their best squared distance so far)
//   rows 1 + cap .. 2 cap the same for ground truth -> map;  unused rows carry the bound -1 (a walk that ends at the root)
// all-gathered as it is (world x (1 + 2 cap) x 4 doubles), answered in place, min-reduced, patched.
// ------------------------------------------------------------------------------------------------------------
__global__ void k_cross_message(const SPoint *__restrict__ sp0, const unsigned int *__restrict__ list0, const double *__restrict__ d2s0, long long cnt0,
                                const SPoint *__restrict__ sp1, const unsigned int *__restrict__ list1, const double *__restrict__ d2s1, long long cnt1,
                                long long cap, double n_loc0, double n_loc1, double *__restrict__ msg) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        msg[0] = (double) cnt0;
        msg[1] = (double) cnt1;
        msg[2] = n_loc0;
        msg[3] = n_loc1;
    }
    if (t >= 2 * cap) return;
    const int d = t >= cap ? 1 : 0;
    const long long k = t - d * cap;
    const long long cnt = d ? cnt1 : cnt0;
    double *row = msg + 4 * (1 + t);
    if (k < cnt) {
        const unsigned int at = d ? list1[k] : list0[k];
        const SPoint q = d ? sp1[at] : sp0[at];
        row[0] = q.x;
        row[1] = q.y;
        row[2] = q.z;
        row[3] = d ? d2s1[at] : d2s0[at];
    } else {
        row[0] = row[1] = row[2] = 0.0;
        row[3] = -1.0;
    }
}

struct CrossCuts {
    double c[66];  // cuts[world + 1]
};
// gathered message -> the queries of ONE direction as SPoints, their bounds (own rank / padding: -1) and the band each owner has searched
__global__ void k_cross_expand(const double *__restrict__ gathered, int world, long long cap, int dir, int own_rank, CrossCuts cuts, double halo,
                               SPoint *__restrict__ qs, double *__restrict__ bound, double *__restrict__ cov) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long) world * cap) return;
    const int k = (int) (t / cap);
    const long long j = t - (long long) k * cap;
    const double *row = gathered + 4 * ((long long) k * (1 + 2 * cap) + 1 + (long long) dir * cap + j);
    SPoint p;
    p.x = row[0];
    p.y = row[1];
    p.z = row[2];
    p.idx = t;
    qs[t] = p;
    bound[t] = (k == own_rank) ? -1.0 : row[3];
    cov[2 * t] = cuts.c[k] - halo;       // the owner's slab + halo holds every point of the cloud in this band (me_halo_pack_device)
    cov[2 * t + 1] = cuts.c[k + 1] + halo;
}
// answers back into the (world x (1 + 2 cap)) block: searched slots = min(bound, nearest here), the own rank's slots = its own bounds
__global__ void k_cross_collect(const double *__restrict__ gathered, const double *__restrict__ ans, int world, long long cap, int dir, int own_rank,
                                double *__restrict__ d2_out) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long) world * cap) return;
    const int k = (int) (t / cap);
    const long long j = t - (long long) k * cap;
    const long long at = (long long) k * (1 + 2 * cap) + 1 + (long long) dir * cap + j;
    d2_out[at] = (k == own_rank) ? gathered[4 * at + 3] : ans[t];
    if (j == 0 && dir == 0) d2_out[(long long) k * (1 + 2 * cap)] = 0.0;  // the header slot: never read back, but it goes through a min-reduce
}
__global__ void k_cross_untouched(const double *__restrict__ gathered, int world, long long cap, int dir, double *__restrict__ d2_out) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long) world * cap) return;
    const int k = (int) (t / cap);
    const long long at = (long long) k * (1 + 2 * cap) + 1 + (long long) dir * cap + (t - (long long) k * cap);
    d2_out[at] = gathered[4 * at + 3];
    if (t == (long long) k * cap && dir == 0) d2_out[(long long) k * (1 + 2 * cap)] = 0.0;  // (header slot, as k_cross_collect)
}
__global__ void k_cross_patch(const unsigned int *__restrict__ list0, long long cnt0, double *__restrict__ d2_0, const unsigned int *__restrict__ list1,
                              long long cnt1, double *__restrict__ d2_1, const double *__restrict__ mine, long long cap) {
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt0) {
        const unsigned int i = list0[t];
        d2_0[i] = fmin(d2_0[i], mine[1 + t]);
    } else if (t - cnt0 < cnt1) {
        const long long u = t - cnt0;
        const unsigned int i = list1[u];
        d2_1[i] = fmin(d2_1[i], mine[1 + cap + u]);
    }
}

int nn_cross_message(me_ctx *ctx, double *msg_device, long long cap, long long n_loc_est, long long n_loc_gt, long long counts[2]) {
    if (!msg_device || cap < 1 || !counts) return ctx->fail(ME_ERR_ARG, "me_nn_cross_message: bad argument");
    Cloud &a = ctx->cloud[ME_SLOT_EST], &b = ctx->cloud[ME_SLOT_GT];
    if (a.nn_ref_slot < 0 || b.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "me_nn_cross_message: search both directions first (me_nn1)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    counts[0] = a.n_unres;
    counts[1] = b.n_unres;
    const long long c0 = std::min(a.n_unres, cap), c1 = std::min(b.n_unres, cap);  // (more than cap: the caller's exact-size fallback)
    hipLaunchKernelGGL(k_cross_message, dim3((unsigned int) ((2 * cap + 255) / 256)), dim3(256), 0, ctx->stream, a.sp.as<SPoint>(),
                       a.nn_unres.as<unsigned int>(), a.nn_d2.as<double>(), c0, b.sp.as<SPoint>(), b.nn_unres.as<unsigned int>(),
                       b.nn_d2.as<double>(), c1, cap, (double) n_loc_est, (double) n_loc_gt, msg_device);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int nn_cross_answer(me_ctx *ctx, const double *gathered_device, int world, long long cap, int own_rank, int dir_mask, int axis,
                    const double *cuts_host, double halo, double *d2_device) {
    if (!gathered_device || !d2_device || !cuts_host || world < 1 || world > 64 || cap < 1 || own_rank < 0 || own_rank >= world || axis < 0 ||
        axis > 2)
        return ctx->fail(ME_ERR_ARG, "me_nn_cross_answer: bad argument");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    for (int slot = 0; slot < 2; ++slot) ME_TRY(cloud_finish_octree(ctx, slot));
    CrossCuts cc;
    for (int k = 0; k <= world; ++k) cc.c[k] = cuts_host[k];
    const long long m = (long long) world * cap;
    const unsigned int nb = (unsigned int) ((m + 255) / 256);
    DevBuf &qs = ctx->tmp[0], &qi = ctx->tmp[1], &bd = ctx->tmp[2], &cv = ctx->tmp[3];
    ME_CHECK(ctx, qs.ensure((size_t) m * sizeof(SPoint)));
    ME_CHECK(ctx, qi.ensure((size_t) m * 4));
    ME_CHECK(ctx, bd.ensure((size_t) m * 8));
    ME_CHECK(ctx, cv.ensure((size_t) m * 16));
    ME_CHECK(ctx, ctx->nn_far.ensure((size_t) m * 4 + 64));
    ME_CHECK(ctx, ctx->red.ensure(64));
    unsigned int *d_cnt = ctx->red.as<unsigned int>();
    for (int dir = 0; dir < 2; ++dir) {
        if (!((dir_mask >> dir) & 1)) {  // nobody else has an open query in this direction: every slot keeps its owner's bound
            hipLaunchKernelGGL(k_cross_untouched, dim3(nb), dim3(256), 0, ctx->stream, gathered_device, world, cap, dir, d2_device);
            continue;
        }
        Cloud &r = ctx->cloud[dir == 0 ? ME_SLOT_GT : ME_SLOT_EST];  // the queries of the map are answered from the ground truth held here
        if (!r.uploaded) return ctx->fail(ME_ERR_STATE, "me_nn_cross_answer: reference cloud not uploaded");
        hipLaunchKernelGGL(k_cross_expand, dim3(nb), dim3(256), 0, ctx->stream, gathered_device, world, cap, dir, own_rank, cc, halo,
                           qs.as<SPoint>(), bd.as<double>(), cv.as<double>());
        if (r.n > 0) {
            ME_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 32, ctx->stream));
            TimerScope ts(ctx, "nn1_cross");  // (the other ranks' open queries: k_nn1 + k_nn_far)
            unsigned long long *dbgp = ctx->timers_on ? ctx->nn1_dbg() : nullptr;
            const dim3 g1((unsigned int) std::min<long long>((m + 7) / 8, 256 * 4 * kNn1Waves));
            if (dbgp)
                hipLaunchKernelGGL((k_nn1<true, true>), g1, dim3(kNn1Block), nn1_cache_bytes(r.oct), ctx->stream, qs.as<SPoint>(), 0LL, m, r.sp.as<SPoint>(),
                                   r.n, r.oct, bd.as<double>(), qi.as<int>(), (const unsigned int *) nullptr, (const unsigned int *) nullptr, 1, dbgp,
                                   ctx->nn_far.as<unsigned int>(), d_cnt + 1, nn1_far_cap(), d_cnt + 4, cv.as<double>(), axis);
            else
                hipLaunchKernelGGL((k_nn1<true, false>), g1, dim3(kNn1Block), nn1_cache_bytes(r.oct), ctx->stream, qs.as<SPoint>(), 0LL, m, r.sp.as<SPoint>(),
                                   r.n, r.oct, bd.as<double>(), qi.as<int>(), (const unsigned int *) nullptr, (const unsigned int *) nullptr, 1,
                                   (unsigned long long *) nullptr, ctx->nn_far.as<unsigned int>(), d_cnt + 1, nn1_far_cap(), d_cnt + 4, cv.as<double>(), axis);
            hipLaunchKernelGGL(k_nn_far, dim3(kFarGrid), dim3(256), 0, ctx->stream, qs.as<SPoint>(), 0LL, r.sp.as<SPoint>(), r.oct, bd.as<double>(),
                               qi.as<int>(), ctx->nn_far.as<unsigned int>(), d_cnt + 1, nn_far_leaf(), dbgp, cv.as<double>(), axis);
        }
        hipLaunchKernelGGL(k_cross_collect, dim3(nb), dim3(256), 0, ctx->stream, gathered_device, bd.as<double>(), world, cap, dir, own_rank, d2_device);
    }
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

int nn_cross_patch(me_ctx *ctx, const double *d2_reduced_device, long long cap, int own_rank) {
    if (!d2_reduced_device || cap < 1 || own_rank < 0) return ctx->fail(ME_ERR_ARG, "me_nn_cross_patch: bad argument");
    Cloud &a = ctx->cloud[ME_SLOT_EST], &b = ctx->cloud[ME_SLOT_GT];
    const long long c0 = std::min(a.n_unres, cap), c1 = std::min(b.n_unres, cap);
    if (c0 + c1 == 0) return ME_OK;
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_cross_patch, dim3((unsigned int) ((c0 + c1 + 255) / 256)), dim3(256), 0, ctx->stream, a.nn_unres.as<unsigned int>(), c0,
                       a.nn_d2.as<double>(), b.nn_unres.as<unsigned int>(), c1, b.nn_d2.as<double>(),
                       d2_reduced_device + (long long) own_rank * (1 + 2 * cap), cap);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int nn_patch(me_ctx *ctx, int qslot, const double *d2_device, long long count) {
    if (qslot < 0 || qslot > 1 || (count > 0 && !d2_device)) return ctx->fail(ME_ERR_ARG, "me_nn_patch: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (count != q.n_unres) return ctx->fail(ME_ERR_ARG, "me_nn_patch: count differs from me_nn_unresolved");
    if (count == 0) return ME_OK;
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_nn_patch, dim3((unsigned int) ((count + 255) / 256)), dim3(256), 0, ctx->stream,
                       q.nn_unres.as<unsigned int>(), count, d2_device, q.nn_d2.as<double>());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int nn_fetch(me_ctx *ctx, int qslot, int32_t *idx, double *d2) {
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "no NN result for this slot (call me_nn1 first)");
    if (!idx && !d2) return ME_OK;
    long long b, e;
    ctx->shard_range(q.n, b, e);
    DevBuf &od = ctx->tmp[0], &oi = ctx->tmp[1];
    ME_CHECK(ctx, od.ensure((size_t) q.n * 8));
    ME_CHECK(ctx, oi.ensure((size_t) q.n * 4));
    if (ctx->shard_world > 1) {  // entries outside the shard read as 0 / -1
        ME_CHECK(ctx, hipMemsetAsync(od.p, 0, (size_t) q.n * 8, ctx->stream));
        ME_CHECK(ctx, hipMemsetAsync(oi.p, 0xFF, (size_t) q.n * 4, ctx->stream));
    }
    if (e > b)
        hipLaunchKernelGGL(k_nn_unpermute, dim3((unsigned int) ((e - b + 255) / 256)), dim3(256), 0, ctx->stream,
                           q.sp.as<SPoint>(), b, e, q.nn_d2.as<double>(), q.nn_idx.as<int>(), od.as<double>(), oi.as<int>());
    if (d2) ME_TRY(copy_d2h(ctx, d2, od.p, (size_t) q.n * 8));
    if (idx) ME_TRY(copy_d2h(ctx, idx, oi.p, (size_t) q.n * 4));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int nn_partial(me_ctx *ctx, int qslot, double gate, int gate_mode, const double trunc[5], me_nn_partial *out) {
    ME_TRACE_POINT(ctx, "nn_partial: enter");
    if (qslot < 0 || qslot > 1 || !out || !trunc) return ctx->fail(ME_ERR_ARG, "me_nn_partial_sums: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "no NN result for this slot (call me_nn1 first)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    long long b, e;
    ctx->shard_range(q.n, b, e);
    const StatParams sp = make_params(gate, gate_mode, trunc);
    const int nb = (int) std::max<long long>(1, std::min<long long>(1024, (e - b + 255) / 256));
    const size_t bytes_d = (size_t) (nb + 1) * kStatD * 8, bytes_i = (size_t) (nb + 1) * kStatI * 8;
    ME_CHECK(ctx, ctx->red.ensure(bytes_d + bytes_i));
    double *pd = ctx->red.as<double>();
    long long *pi = reinterpret_cast<long long *>(ctx->red.as<char>() + bytes_d);
    {
        TimerScope ts(ctx, "nn_stats");
        hipLaunchKernelGGL(k_nn_partial, dim3(nb), dim3(256), 0, ctx->stream, q.nn_d2.as<double>(), b, e, sp, pd, pi);
        hipLaunchKernelGGL(k_final_sum_d, dim3(kStatD), dim3(256), 0, ctx->stream, pd, nb, kStatD, pd + (size_t) nb * kStatD);
        hipLaunchKernelGGL(k_final_sum_i, dim3(kStatI), dim3(256), 0, ctx->stream, pi, nb, kStatI, pi + (size_t) nb * kStatI);
    }
    double hd[kStatD];
    long long hi[kStatI];
    {
        MailGuard mg(ctx);  // (hd / hi are locals)
        ME_TRY(mail_post(ctx, hd, pd + (size_t) nb * kStatD, sizeof(hd)));
        ME_TRY(mail_post(ctx, hi, pi + (size_t) nb * kStatI, sizeof(hi)));
        ME_TRY(mg.sync());
    }
    out->n_query = e - b;
    out->n_corr = hi[0];
    for (int k = 0; k < 5; ++k) {
        out->n_inl[k] = hi[1 + k];
        out->sum_d[k] = hd[k];
        out->sum_d2[k] = hd[5 + k];
    }
    out->sum_sqrt_all = hd[10];
    return ME_OK;
}

int nn_sigma(me_ctx *ctx, int qslot, double gate, int gate_mode, const double mean[5], double sigma_num[5]) {
    if (qslot < 0 || qslot > 1 || !mean || !sigma_num) return ctx->fail(ME_ERR_ARG, "me_nn_sigma_sums: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "no NN result for this slot (call me_nn1 first)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    long long b, e;
    ctx->shard_range(q.n, b, e);
    const double zeros[5] = {0, 0, 0, 0, 0};
    const StatParams sp = make_params(gate, gate_mode, zeros);
    Mean5 m;
    for (int k = 0; k < 5; ++k) m.m[k] = mean[k];
    const int nb = (int) std::max<long long>(1, std::min<long long>(1024, (e - b + 255) / 256));
    ME_CHECK(ctx, ctx->red.ensure((size_t) (nb + 1) * 5 * 8));
    double *pd = ctx->red.as<double>();
    {
        TimerScope ts(ctx, "nn_stats");
        hipLaunchKernelGGL(k_nn_sigma, dim3(nb), dim3(256), 0, ctx->stream, q.nn_d2.as<double>(), b, e, sp, m, pd);
        hipLaunchKernelGGL(k_final_sum_d, dim3(5), dim3(256), 0, ctx->stream, pd, nb, 5, pd + (size_t) nb * 5);
    }
    {
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, sigma_num, pd + (size_t) nb * 5, 5 * 8));
        ME_TRY(mg.sync());
    }
    return ME_OK;
}

// ---- point-to-point ICP: correspondence + reduction step (Open3D TransformationEstimationPointToPoint) ----
constexpr int kIcpD = 16;  // sum_p[3], sum_q[3], sum_pq[9], sum_d2

__global__ void __launch_bounds__(256)
k_icp_p2p(const SPoint *__restrict__ qsp, const double *__restrict__ d2s, const int *__restrict__ idxs,
          const double *__restrict__ ref_xyz, long long n, double gate2, double cx, double cy, double cz,
          double *__restrict__ pd, long long *__restrict__ pc) {
    double s[kIcpD];
#pragma unroll
    for (int k = 0; k < kIcpD; ++k) s[k] = 0;
    long long cnt = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const double d2 = d2s[i];
        if (d2 >= 0.0 && d2 < gate2) {  // SearchHybrid(q, max, 1): d2 < max^2 [Open3D, upstream]
            const long long j = idxs[i];
            // coordinates relative to a common origin: keeps sum(p q^T) - n pbar qbar^T well conditioned
            const double px = qsp[i].x - cx, py = qsp[i].y - cy, pz = qsp[i].z - cz;
            const double qx = ref_xyz[3 * j] - cx, qy = ref_xyz[3 * j + 1] - cy, qz = ref_xyz[3 * j + 2] - cz;
            s[0] += px; s[1] += py; s[2] += pz;
            s[3] += qx; s[4] += qy; s[5] += qz;
            s[6] = fma(px, qx, s[6]); s[7] = fma(px, qy, s[7]); s[8] = fma(px, qz, s[8]);
            s[9] = fma(py, qx, s[9]); s[10] = fma(py, qy, s[10]); s[11] = fma(py, qz, s[11]);
            s[12] = fma(pz, qx, s[12]); s[13] = fma(pz, qy, s[13]); s[14] = fma(pz, qz, s[14]);
            s[15] += d2;
            ++cnt;
        }
    }
    __shared__ double smd[4];
    __shared__ long long smi[4];
#pragma unroll
    for (int k = 0; k < kIcpD; ++k) {
        const double r = block_sum_256(s[k], smd);
        if (threadIdx.x == 0) pd[(long long) blockIdx.x * kIcpD + k] = r;
    }
    const long long rc = block_sum_256_ll(cnt, smi);
    if (threadIdx.x == 0) pc[blockIdx.x] = rc;
}

int icp_p2p_sums(me_ctx *ctx, int qslot, double max_distance, me_icp_sums *out) {
    if (qslot < 0 || qslot > 1 || !out || !(max_distance > 0)) return ctx->fail(ME_ERR_ARG, "me_icp_p2p_sums: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "me_icp_p2p_sums: call me_nn1(query_slot, ref_slot) first");
    if (q.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_icp_p2p_sums: not available in slab mode");
    Cloud &r = ctx->cloud[q.nn_ref_slot];
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    // me_set_shard: this rank's share [b, e) of the sorted queries (the one me_nn1 searched); the caller all-reduces the sums
    long long sb, se;
    ctx->shard_range(q.n, sb, se);
    const long long n = se - sb;
    const int nb = (int) std::max<long long>(1, std::min<long long>(1024, (n + 255) / 256));
    const size_t bytes_d = (size_t) (nb + 1) * kIcpD * 8;
    ME_CHECK(ctx, ctx->red.ensure(bytes_d + (size_t) (nb + 1) * 8));
    double *pd = ctx->red.as<double>();
    long long *pc = reinterpret_cast<long long *>(ctx->red.as<char>() + bytes_d);
    const double cx = r.origin[0], cy = r.origin[1], cz = r.origin[2];
    {
        TimerScope ts(ctx, "icp");
        hipLaunchKernelGGL(k_icp_p2p, dim3(nb), dim3(256), 0, ctx->stream, q.sp.as<SPoint>() + sb, q.nn_d2.as<double>() + sb,
                           q.nn_idx.as<int>() + sb, r.xyz.as<double>(), n, max_distance * max_distance, cx, cy, cz, pd, pc);
        hipLaunchKernelGGL(k_final_sum_d, dim3(kIcpD), dim3(256), 0, ctx->stream, pd, nb, kIcpD, pd + (size_t) nb * kIcpD);
        hipLaunchKernelGGL(k_final_sum_i, dim3(1), dim3(256), 0, ctx->stream, pc, nb, 1, pc + nb);
    }
    double hd[kIcpD];
    long long hc = 0;
    ME_TRY(copy_d2h(ctx, hd, pd + (size_t) nb * kIcpD, sizeof(hd)));
    ME_TRY(copy_d2h(ctx, &hc, pc + nb, 8));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    out->n_corr = hc;
    out->n_source = q.n;  // (the whole source cloud, also under me_set_shard: fitness = all-reduced n_corr / n_source)
    for (int k = 0; k < 3; ++k) {
        out->sum_p[k] = hd[k];
        out->sum_q[k] = hd[3 + k];
        out->origin[k] = r.origin[k];
    }
    for (int k = 0; k < 9; ++k) out->sum_pq[k] = hd[6 + k];
    out->sum_d2 = hd[15];
    return ME_OK;
}

}  // namespace me
