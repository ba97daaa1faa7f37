// me_mme7.hip — Mean Map Entropy with the neighbourhood moments on the int8 MATRIX pipe (round 4).
// Same result as k_mme3 (me_mme.hip): radius neighbourhood of every point -> covariance -> 0.5 ln(2 pi e det)
// (ComputeMeanMapEntropyUsingNormalTBB map_eval.cpp:1608-1737, the OpenMP / serial variants :1538-1606, :1438-1535).
//
// Why.  k_mme3 spends ~60 % of its time adding (p - q), (p - q)(p - q)^T of the accepted candidates in fp64 on the vector
// unit: 13 instructions for every candidate SOME lane of the wave accepts, at ~20 % lane efficiency — the SIMT price of 64
// queries sharing one candidate stream.  A sum over accepted candidates is a contraction over candidates,
//        D[feature][query] += F[feature][candidate] * accept[candidate][query],
// and the matrix pipe does not care how few ones a mask column has.  The fp64 and f32 MFMAs of gfx950 run at the vector rate
// and block the vector unit (profiles/r04_issue_rates.txt: wave-specialised f32 MFMA || v_fma: 0 % overlap); the int8 MFMA does
// neither (16.3 cycles per 16x16x64, 67 % of it hidden beside vector work).  So the features are exact INTEGERS in signed
// base-256 digits (me_mme_fx.hpp: 80 digit columns per point, built once per cloud by k_mme_feat), the mask is 0 / -1, and
// v_mfma_i32_16x16x64_i8 adds the digit columns in int32 without rounding.  The squared-distance pre-test of k_mme3
// (u = |p'|^2 + a.p' against r^2 - |q'|^2 in FP32, exact fp64 test for the pairs inside the error band) becomes one
// v_mfma_f32_16x16x4_f32 per 16 candidates x 16 queries, with -(T + E) as the C operand so that the SIGN of the result is the
// accept bit and v_perm_b32's sign-select bytes turn four results into four mask bytes in three instructions.
//
// Layout of a step (64 candidates x 64 queries per wavefront):
//   * a wavefront owns 64 curve-consecutive queries (lane l <-> query l while the run table is built and in the epilogue);
//     for the matrix work lane l = 16 g + j stands for query j of each of FOUR query groups G = 0..3 (queries 16 G + j) and for
//     lane-group g of the MFMA operands;
//   * candidates come in CHUNKS of 16 (a cell's points, padded: the feature store is per cell), four chunks t = 0..3 per step;
//   * distance MFMA (per G, per t): A[m][k] = component k of (p'x, p'y, p'z, |p'|^2) of candidate m, B[k][n] = component k of
//     (ax, ay, az, 1) of query n, C = -(T_n + E): lane (n, g) gets x for candidates 4 g + r, r = 0..3, of chunk t;
//   * mask operand of lane (n, g): byte 4 t + r = (x < 0) ? -1 : 0 — K slot (g, 4 t + r) <-> candidate 4 g + r of chunk t;
//   * feature operand of lane (m, g), tile T: dword t = column 16 T + m of candidates 4 g .. 4 g + 3 of chunk t: ONE aligned
//     dword of the chunk's [80 columns][16 candidates] byte matrix;
//   * moment MFMA (per G, per tile T): D[m][n] -= column 16 T + m summed over the accepted candidates of query n; lane (n, g)
//     holds columns 16 T + 4 g + r.
// After a round the digit sums are put together per query (4 x 4 transposes across the lane groups with the half-exchange
// permutes, then me_mme_fx.hpp's exact 128-bit arithmetic), giving k, sum(p - q), sum((p - q)(p - q)^T) about the query — the
// quantities k_mme3 accumulates in fp64 — and the covariance / determinant / entropy code is k_mme3's.
// Exactness: the accepted set is the fp64 test's (band pairs are decided by ((dx*dx + dy*dy) + dz*dz) < r^2 on the original
// coordinates); counts and valid flags are identical to k_mme3's, entropies agree to ~1e-13 (the sums here are exact, k_mme3's
// carry fp64 rounding).
//
// MEASURED AND NOT ADOPTED (round 4; compiled only with `make EXTRA=-DME_AB`, selected with ME_MME_V=7): results identical to
// k_mme3's (every MME parity test passes with it as the default kernel), 22.2 ms per 50 M-query launch against 12.8.  Per
// wavefront (profiles/r04_mme7_sq_per_wave.json): 4 885 VALU instructions against 9 401 (460 of them MFMAs), 665 SALU against
// 3 775, 100 LDS against 1 409 — the matrix formulation halves the instruction count as intended — but 332 vector-memory reads
// against 84 and a wave life of 119 k cycles of which 75 k are spent in s_waitcnt: 80 accumulator registers + operands leave
// TWO waves per SIMD (k_mme3: eight), the feature operands are 80 bytes per candidate and step against k_mme3's 32-byte points
// (51 GB per launch through L2), and the dependent loads of the run-table probes, which eight waves hide, are exposed.
// profiles/EXPERIMENTS.md "Round 4" has the account and what would have to change (a 40-column feature set, or a device with
// twice the registers per SIMD).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "me_internal.hpp"
#include "me_mme_fx.hpp"

#ifdef ME_AB
namespace me {

typedef int mme_v4i __attribute__((ext_vector_type(4)));
typedef float mme_v4f __attribute__((ext_vector_type(4)));

constexpr int kMme7List = 128;  // chunk-list entries per wavefront (a multiple of 4)

// ---- cross-lane helpers -------------------------------------------------------------------------------------------------
// 4 x 4 transpose between the register index and the lane-group index (lanes 16 G .. 16 G + 15 = group G): afterwards
// register t of group g holds what register g of group t held.  v_permlane32_swap trades the upper half of its first
// operand for the lower half of its second (swaps the high bit of the group with the register pair), v_permlane16_swap the
// odd rows of the first for the even rows of the second (the low bit).
__device__ __forceinline__ void transpose4_u32(unsigned int &r0, unsigned int &r1, unsigned int &r2, unsigned int &r3) {
    auto a = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
    auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    r0 = c[0];
    r1 = c[1];
    r2 = d[0];
    r3 = d[1];
}
__device__ __forceinline__ void transpose4_f32(float &r0, float &r1, float &r2, float &r3) {
    unsigned int a = __float_as_uint(r0), b = __float_as_uint(r1), c = __float_as_uint(r2), d = __float_as_uint(r3);
    transpose4_u32(a, b, c, d);
    r0 = __uint_as_float(a);
    r1 = __uint_as_float(b);
    r2 = __uint_as_float(c);
    r3 = __uint_as_float(d);
}
__device__ __forceinline__ void transpose4_i64(long long &r0, long long &r1, long long &r2, long long &r3) {
    unsigned int l0 = (unsigned int) r0, l1 = (unsigned int) r1, l2 = (unsigned int) r2, l3 = (unsigned int) r3;
    unsigned int h0 = (unsigned int) ((unsigned long long) r0 >> 32), h1 = (unsigned int) ((unsigned long long) r1 >> 32),
                 h2 = (unsigned int) ((unsigned long long) r2 >> 32), h3 = (unsigned int) ((unsigned long long) r3 >> 32);
    transpose4_u32(l0, l1, l2, l3);
    transpose4_u32(h0, h1, h2, h3);
    r0 = (long long) (((unsigned long long) h0 << 32) | l0);
    r1 = (long long) (((unsigned long long) h1 << 32) | l1);
    r2 = (long long) (((unsigned long long) h2 << 32) | l2);
    r3 = (long long) (((unsigned long long) h3 << 32) | l3);
}
// four mask bytes from the signs of four floats: byte r = (x_r < 0 or -0) ? 0xff : 0x00.  v_perm_b32 selector 9 / 11 = "0xff if
// bit 31 of the second / first source is set"; 12 = 0x00.
__device__ __forceinline__ unsigned int sign_bytes(float x0, float x1, float x2, float x3) {
    const unsigned int lo = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x0c0c0b09u);
    const unsigned int hi = __builtin_amdgcn_perm(__float_as_uint(x3), __float_as_uint(x2), 0x0b090c0cu);
    return lo | hi;
}
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// ---- feature store ----------------------------------------------------------------------------------------------------------
// chunks per cell (+ the largest cell population, for the scale / overflow guard)
__global__ void k_mme_chunks(const unsigned int *__restrict__ cell_start, long long n_cells, unsigned int *__restrict__ nch,
                             unsigned int *__restrict__ pmax) {
    const long long ci = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int cnt = 0;
    if (ci < n_cells) {
        cnt = cell_start[ci + 1] - cell_start[ci];
        nch[ci] = (cnt + fx::kChunk - 1) / fx::kChunk;
    } else if (ci == n_cells) {
        nch[ci] = 0;  // (the scan runs over n_cells + 1 entries: the last one becomes the total)
    }
    for (int o = 32; o > 0; o >>= 1) cnt = max(cnt, (unsigned int) __shfl_xor((int) cnt, o, 64));
    if ((threadIdx.x & 63) == 0 && cnt) atomicMax(pmax, cnt);
}
// (first point, valid count) of every chunk
__global__ void k_mme_chunk_desc(const unsigned int *__restrict__ cell_start, const unsigned int *__restrict__ cell_chunk,
                                 long long n_cells, uint2 *__restrict__ desc) {
    const long long ci = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n_cells) return;
    const unsigned int cs = cell_start[ci], cnt = cell_start[ci + 1] - cs;
    const unsigned int cb = cell_chunk[ci];
    for (unsigned int i = 0; i * fx::kChunk < cnt; ++i)
        desc[cb + i] = make_uint2(cs + i * fx::kChunk, min((unsigned int) fx::kChunk, cnt - i * fx::kChunk));
}
// One lane per padded candidate slot; a wavefront = four chunks.  The lane computes its point's 80 digit bytes, the wave
// transposes them through LDS into the chunk's [column][candidate] byte matrix and writes it with coalesced dwords.
__global__ void __launch_bounds__(256)
k_mme_feat(const SPoint *__restrict__ sp, const uint2 *__restrict__ desc, long long n_chunks, fx::Frame f,
           unsigned int *__restrict__ feat) {
    __shared__ __attribute__((aligned(16))) unsigned char s_t[4][4 * fx::kChunkBytes];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned char *t = s_t[wv];
    const long long chunk0 = ((long long) blockIdx.x * 4 + wv) * 4;  // the wave's first chunk
    const int ct = lane >> 4, m = lane & 15;
    const long long chunk = chunk0 + ct;
    unsigned char b[fx::kCols];
#pragma unroll
    for (int i = 0; i < fx::kCols; ++i) b[i] = 0;
    if (chunk < n_chunks) {
        const uint2 d = desc[chunk];
        if ((unsigned int) m < d.y) {
            const SPoint p = sp[d.x + m];
            const unsigned long long X = (unsigned long long) (fx::fix(p.x, f.s) - f.ox), Y = (unsigned long long) (fx::fix(p.y, f.s) - f.oy),
                                     Z = (unsigned long long) (fx::fix(p.z, f.s) - f.oz);
            fx::point_features(X, Y, Z, b);
        }
    }
    unsigned char *tc = t + ct * fx::kChunkBytes;
#pragma unroll
    for (int c = 0; c < fx::kCols; ++c) tc[c * fx::kChunk + m] = b[c];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the wave's four chunks are contiguous in the store: 4 x 1280 bytes = 1280 dwords, 20 per lane
    const unsigned int *tw = reinterpret_cast<const unsigned int *>(t);
    const long long left = n_chunks - chunk0;
    const int ndw = (int) (left >= 4 ? 4 : (left > 0 ? left : 0)) * (fx::kChunkBytes / 4);
    unsigned int *out = feat + chunk0 * (fx::kChunkBytes / 4);
    for (int i = lane; i < ndw; i += 64) out[i] = tw[i];
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
// int64 from four digit sums: c0 + c1 2^8 + c2 2^16 + c3 2^24
__device__ __forceinline__ long long comb4(int c0, int c1, int c2, int c3) {
    return (long long) c0 + ((long long) c1 << 8) + ((long long) c2 << 16) + ((long long) c3 << 24);
}
__device__ __forceinline__ long long pack2(int lo, int hi) {
    return (long long) (((unsigned long long) (unsigned int) hi << 32) | (unsigned int) lo);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_mme7(const SPoint *__restrict__ sp, const unsigned long long *__restrict__ codes, long long i_begin, long long i_end,
       GridView g, const unsigned int *__restrict__ cell_chunk, const unsigned char *__restrict__ feat, FrameView fr,
       fx::Frame fxf, SlabView slab, double r2, int min_k, double *__restrict__ ent_s, unsigned char *__restrict__ valid_s,
       double *__restrict__ part_sum, long long *__restrict__ part_cnt, unsigned int xcd_chunk, double cell_h, float thr_hi,
       unsigned int band_bits) {
    // thr_hi = r^2 + E in FP32, band_bits = the bit pattern of 2 E (E = 2^-12 cell_h^2 bounds everything FP32 does to
    // x = u - (T + E); derivation at k_mme3, re-done below for the box CENTRE as origin and the MFMA's four-step fmaf chain)
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
    __shared__ unsigned int s_tabc[4][kGroupTab + 1];
    __shared__ unsigned int s_rows[4][kGroupRows];
    __shared__ uint2 s_list[4][kMme7List];
    __shared__ float4 s_qa[4][64];
    __shared__ double s_qd[4][3][64];
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    int2 *tab = s_tab[wv];
    unsigned int *tabc = s_tabc[wv];
    uint2 *list = s_list[wv];
    float4 *qa = s_qa[wv];
    const int lane = threadIdx.x & 63;
    const int lj = lane & 15, lg = lane >> 4;
    // the query's coordinates for the exact test of band pairs (rare path), read by the lanes that stand for it
    s_qd[wv][0][lane] = qx;
    s_qd[wv][1][lane] = qy;
    s_qd[wv][2][lane] = qz;
    const int band_thr = (int) (0x80000000u + band_bits);  // as a signed int: x in [-2E, -0]  <=>  bits(x) <= band_thr

    double det_keep = 0.0;
    bool have_det = false;
    while (__ballot(!done)) {
        GroupBox bx;
        int nk = 0;
        const int cx = (int) compact21(mycell), cy = (int) compact21(mycell >> 1), cz = (int) compact21(mycell >> 2);
        const bool in = wave_group_table<1, true, kGroupR, true>(!done, cx, cy, cz, g, cell_lim, lane, tab, bx, &nk, s_rows[wv], nullptr, tabc,
                                                  cell_chunk);
        // origin = CENTRE of the round's cell box (<= 7 cells per axis): |p'|, |q'| <= 3.5 h per axis, |a| <= 7 h.  Error of
        // x = fmaf chain (C, then the three products, then |p'|^2) against d^2 - r^2 - E, in units of 2^-24 h^2: rounding of
        // p' and a: 3 * 2 * (3.5 * 7) = 147;  |p'|^2 <= 36.75: 37;  four fmaf steps, partial sums <= 38 + 73.5 + 36.75 < 150:
        // 600;  |a|^2 / 4 (<= 36.75, three roundings): 111;  T's rounding: 38;  r^2 + E in FP32: 1.   Sum 934 < 4096 = E.
        const int nz = nk / (bx.nx * bx.ny);
        const double ox = uniform_f64(fr.ox + ((double) bx.x0 + 0.5 * (double) bx.nx) * cell_h),
                     oy = uniform_f64(fr.oy + ((double) bx.y0 + 0.5 * (double) bx.ny) * cell_h),
                     oz = uniform_f64(fr.oz + ((double) bx.z0 + 0.5 * (double) nz) * cell_h);
        {
            const float ax = (float) (-2.0 * (qx - ox)), ay = (float) (-2.0 * (qy - oy)), az = (float) (-2.0 * (qz - oz));
            const float s = fmaf(az, az, fmaf(ay, ay, ax * ax));
            // (the group predicate rides on the threshold: lanes outside the group accept nothing)
            const float t_hi = in ? fmaf(-0.25f, s, thr_hi) : -INFINITY;
            qa[lane] = make_float4(ax, ay, az, t_hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float bq[4], cneg[4];  // per query group: B operand of the distance MFMA (component lg of (a, 1)), C operand -(T + E)
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const float4 v = qa[16 * G + lj];
            bq[G] = lg == 0 ? v.x : (lg == 1 ? v.y : (lg == 2 ? v.z : 1.0f));
            cneg[G] = -v.w;
        }
        mme_v4i acc[4][fx::kTiles];
#pragma unroll
        for (int G = 0; G < 4; ++G)
#pragma unroll
            for (int T = 0; T < fx::kTiles; ++T) acc[G][T] = mme_v4i{0, 0, 0, 0};

        // one step per four list entries.  The loads of step i + 1 (the chunk's points, its feature dwords) are issued before the
        // matrix work of step i: with two resident waves per SIMD nothing else hides their latency (rocprofv3, first version of
        // this kernel: 63 % of a wave's cycles in s_waitcnt).
        auto process = [&](int nent) {
            // (the point of step i + 1 is loaded during step i: with two resident waves per SIMD nothing else hides the latency
            // of the load the whole step waits on; prefetching the 20 feature dwords as well costs 20 more registers and spills)
            uint2 ent_n = list[lg];  // (first point, chunk << 5 | valid count) of chunk t = lg
            double nx = 0, ny = 0, nz = 0;
            if ((unsigned int) lj < (ent_n.y & 31u)) {
                const SPoint p = sp[ent_n.x + lj];
                nx = p.x;
                ny = p.y;
                nz = p.z;
            }
            for (int it = 0; it < nent; it += 4) {
                const uint2 ent = ent_n;
                // feature operands: dword t of tile T = column 16 T + lj, candidates 4 lg .. 4 lg + 3 of chunk t
                mme_v4i af[fx::kTiles];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned int cid = (unsigned int) __builtin_amdgcn_readlane((int) (ent.y >> 5), 16 * t);
                    const unsigned int *fb = reinterpret_cast<const unsigned int *>(feat + (size_t) cid * fx::kChunkBytes) + (lj * 4 + lg);
#pragma unroll
                    for (int T = 0; T < fx::kTiles; ++T) af[T][t] = (int) fb[T * 64];
                }
                float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = INFINITY;  // a padded slot: rank +inf, never accepted
                if ((unsigned int) lj < (ent.y & 31u)) {
                    const double px = nx - ox, py = ny - oy, pz = nz - oz;
                    c0 = (float) px;
                    c1 = (float) py;
                    c2 = (float) pz;
                    // |p'|^2 of the ROUNDED coordinates (the error bound is stated for them), rounded once
                    c3 = (float) (((double) c0 * (double) c0 + (double) c1 * (double) c1) + (double) c2 * (double) c2);
                }
                if (it + 4 < nent) {
                    ent_n = list[it + 4 + lg];
                    if ((unsigned int) lj < (ent_n.y & 31u)) {
                        const SPoint p = sp[ent_n.x + lj];
                        nx = p.x;
                        ny = p.y;
                        nz = p.z;
                    }
                }
                // lane (t, m) holds the four components of candidate m of chunk t; the distance MFMA of chunk t wants component
                // lg of candidate lj in lane (lj, lg): a 4 x 4 transpose across the lane groups
                transpose4_f32(c0, c1, c2, c3);
                const float arec[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    mme_v4f d[4];
                    const mme_v4f cc = {cneg[G], cneg[G], cneg[G], cneg[G]};
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(arec[t], bq[G], cc, 0, 0, 0);
                    // the accepted value closest to the threshold, as a signed int: negative floats order by decreasing magnitude
                    int em = 0x7fffffff;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        em = min(min(em, min(__float_as_int(d[t][0]), __float_as_int(d[t][1]))),
                                 min(__float_as_int(d[t][2]), __float_as_int(d[t][3])));
                    if (__builtin_expect(__ballot(em <= band_thr) != 0ULL, 0)) {
                        // some pair of this group sits in the band x in [-2E, 0]: the fp64 test of the reference decides it
                        asm volatile("; band: exact test" ::: "memory");
                        const double gx = s_qd[wv][0][16 * G + lj], gy = s_qd[wv][1][16 * G + lj], gz = s_qd[wv][2][16 * G + lj];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const unsigned int st = (unsigned int) __builtin_amdgcn_readlane((int) ent.x, 16 * t);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (__float_as_uint(d[t][r]) - 0x80000000u <= band_bits) {
                                    const SPoint p = sp[st + 4 * lg + r];  // (a band value is finite: the slot is a real point)
                                    const double ex = p.x - gx, ey = p.y - gy, ez = p.z - gz;
                                    const double d2 = (ex * ex + ey * ey) + ez * ez;
                                    d[t][r] = d2 < r2 ? -1.0f : 1.0f;  // strict, nanoflann RadiusResultSet [upstream]
                                }
                        }
                    }
                    mme_v4i mk;
#pragma unroll
                    for (int t = 0; t < 4; ++t) mk[t] = (int) sign_bytes(d[t][0], d[t][1], d[t][2], d[t][3]);
#pragma unroll
                    for (int T = 0; T < fx::kTiles; ++T) acc[G][T] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[T], mk, acc[G][T], 0, 0, 0);
                }
            }
        };

        // chunk list: the runs of the table, cut into 16-candidate chunks, kMme7List entries at a time
        int fill = 0;
        for (int base = 0; base < nk; base += 64) {
            const int tsl = base + lane;
            int rs = 0, rc = 0;
            unsigned int cb = 0;
            if (tsl < nk) {
                const int2 run = tab[tsl];
                rs = run.x;
                rc = run.y;
                cb = tabc[tsl];
            }
            while (__ballot(rc > 0)) {
                const int nch = (rc + fx::kChunk - 1) / fx::kChunk;
                const int incl = wave_incl_scan_i(nch, lane);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                const int pos = fill + incl - nch;
                const int emit = max(0, min(nch, kMme7List - pos));
                for (int i = 0; i < emit; ++i)
                    list[pos + i] = make_uint2((unsigned int) (rs + i * fx::kChunk),
                                               ((cb + (unsigned int) i) << 5) | (unsigned int) min(fx::kChunk, rc - i * fx::kChunk));
                rs += emit * fx::kChunk;
                rc = max(0, rc - emit * fx::kChunk);
                cb += (unsigned int) emit;
                fill = min(fill + total, kMme7List);
                if (fill == kMme7List) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    process(kMme7List);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    fill = 0;
                }
            }
        }
        if (fill) {
            const int up = (fill + 3) & ~3;
            if (lane >= fill && lane < up) list[lane] = make_uint2(0u, 0u);  // null chunks: no valid candidate
            if (fill > 64 && lane + 64 >= fill && lane + 64 < up) list[lane + 64] = make_uint2(0u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            process(up);
        }

        // ---- digit sums -> moments about the query.  Lane (lj, lg) holds, for every group G, the columns 16 T + 4 lg + r of
        // query 16 G + lj (negated: the mask is -1).  Four digits -> one int64; then the 4 x 4 transposes bring all 20 partial
        // sums of query `lane` into lane `lane`.
        long long P[fx::kTiles + 1][4];  // [tile (+ one extra for the second half of tile 4)][source lane group after the transpose]
#pragma unroll
        for (int G = 0; G < 4; ++G) {
#pragma unroll
            for (int T = 0; T < 4; ++T) P[T][G] = comb4(-acc[G][T][0], -acc[G][T][1], -acc[G][T][2], -acc[G][T][3]);
            // tile 4: lane groups 0, 1 hold digits 0..7 of V_zz; groups 2, 3 hold eight separate columns (ninth digits, count)
            const int e0 = -acc[G][4][0], e1 = -acc[G][4][1], e2 = -acc[G][4][2], e3 = -acc[G][4][3];
            P[4][G] = lg < 2 ? comb4(e0, e1, e2, e3) : pack2(e0, e1);
            P[5][G] = pack2(e2, e3);
        }
#pragma unroll
        for (int T = 0; T < fx::kTiles + 1; ++T) transpose4_i64(P[T][0], P[T][1], P[T][2], P[T][3]);
        if (in) {
            done = true;
            auto wide = [](long long lo, long long hi) { return (fx::i128) lo + ((fx::i128) hi << 32); };
            const long long k = (long long) (int) (unsigned int) P[5][3];  // column 78
            const int kk = (int) k - 1;  // drop the query itself (map_eval.cpp:1672-1673)
            if (kk >= min_k) {           // (:1675 k >= 10, :1458 k >= 5)
                fx::i128 M[6];
                auto low64 = [](long long lo, long long hi) { return (unsigned long long) lo + ((unsigned long long) hi << 32); };
                const unsigned long long S1[3] = {low64(P[0][0], P[0][1]), low64(P[0][2], P[0][3]), low64(P[1][0], P[1][1])};
                auto ninth = [](long long v, int hi) { return (fx::i128) (int) (unsigned int) (hi ? ((unsigned long long) v >> 32) : (unsigned long long) v) << 64; };
                M[0] = wide(P[1][2], P[1][3]) + ninth(P[4][2], 0);  // xx
                M[1] = wide(P[2][0], P[2][1]) + ninth(P[4][2], 1);  // xy
                M[2] = wide(P[2][2], P[2][3]) + ninth(P[5][2], 0);  // xz
                M[3] = wide(P[3][0], P[3][1]) + ninth(P[5][2], 1);  // yy
                M[4] = wide(P[3][2], P[3][3]) + ninth(P[4][3], 0);  // yz
                M[5] = wide(P[4][0], P[4][1]) + ninth(P[4][3], 1);  // zz
                const unsigned long long Xq = (unsigned long long) (fx::fix(qx, fxf.s) - fxf.ox), Yq = (unsigned long long) (fx::fix(qy, fxf.s) - fxf.oy),
                                         Zq = (unsigned long long) (fx::fix(qz, fxf.s) - fxf.oz);
                const fx::Moments mo = fx::moments_about_query(k, S1, M, Xq, Yq, Zq, fxf.s);
                const double s1x = mo.s1[0], s1y = mo.s1[1], s1z = mo.s1[2];
                const double sxx = mo.s2[0], sxy = mo.s2[1], sxz = mo.s2[2], syy = mo.s2[3], syz = mo.s2[4], szz = mo.s2[5];
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
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
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
        const long long i = i_begin + (long long) loc;
        ent_s[i] = H;  // 0.0 where invalid (:1614)
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

// ---- instruction semantics this file relies on, checked once per process on the device itself --------------------------------
// (operand layouts of the two MFMAs, the half-exchange permutes behind transpose4, v_perm_b32's sign-select bytes)
__global__ void k_mme7_selftest(int *bad) {
    const int l = threadIdx.x, j = l & 15, gq = l >> 4;
    int nbad = 0;
    {   // transpose4: register t of lane group g <- register g of lane group t
        unsigned int r[4];
        for (int i = 0; i < 4; ++i) r[i] = (unsigned int) (l * 4 + i);
        transpose4_u32(r[0], r[1], r[2], r[3]);
        for (int t = 0; t < 4; ++t)
            if (r[t] != (unsigned int) ((16 * t + j) * 4 + gq)) ++nbad;
    }
    {   // sign bytes
        const float x0 = (l & 1) ? -1.5f : 2.0f, x1 = (l & 2) ? -0.0f : 0.0f, x2 = (l & 4) ? -INFINITY : INFINITY, x3 = (l & 8) ? -1e-30f : 1e-30f;
        const unsigned int want = ((l & 1) ? 0xffu : 0u) | ((l & 2) ? 0xff00u : 0u) | ((l & 4) ? 0xff0000u : 0u) | ((l & 8) ? 0xff000000u : 0u);
        if (sign_bytes(x0, x1, x2, x3) != want) ++nbad;
    }
    {   // int8 MFMA: D[m][n] = sum over K slots (g, b) of A[m][(g, b)] B[(g, b)][n]; lane (n, g) register r = D[4 g + r][n]
        auto av = [](int m, int gg, int b) { return (m * 7 + gg * 3 + b) % 5 - 2; };
        auto bv = [](int n, int gg, int b) { return (n * 5 + gg + b * 3) % 3 - 1; };
        mme_v4i a, b;
        for (int w = 0; w < 4; ++w) {
            unsigned int ua = 0, ub = 0;
            for (int e = 0; e < 4; ++e) {
                ua |= (unsigned int) (unsigned char) (signed char) av(j, gq, 4 * w + e) << (8 * e);
                ub |= (unsigned int) (unsigned char) (signed char) bv(j, gq, 4 * w + e) << (8 * e);
            }
            a[w] = (int) ua;
            b[w] = (int) ub;
        }
        mme_v4i c = {1, 2, 3, 4};
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) {
            int want = r + 1;
            for (int gg = 0; gg < 4; ++gg)
                for (int bb = 0; bb < 16; ++bb) want += av(4 * gq + r, gg, bb) * bv(j, gg, bb);
            if (c[r] != want) ++nbad;
        }
    }
    {   // f32 MFMA: an fmaf chain from C over k = 0..3; lane (n, g) register r = D[4 g + r][n]
        const float a = (float) j + 0.25f * (float) gq;  // A[m = j][k = gq]
        const float b = (float) j - (float) gq;          // B[k = gq][n = j]
        mme_v4f c = {1.0f, 1.0f, 1.0f, 1.0f};
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * gq + r;
            float want = 1.0f;
            for (int k = 0; k < 4; ++k) want = fmaf((float) m + 0.25f * (float) k, (float) j - (float) k, want);
            if (c[r] != want) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

static int mme7_selftest(me_ctx *ctx) {
    static int result = -1;  // -1: not run, 0: ok, > 0: mismatches
    if (result >= 0) return result;
    int *d = nullptr, h = -1;
    if (hipMalloc(&d, 4) != hipSuccess) return 1;
    (void) hipMemsetAsync(d, 0, 4, ctx->stream);
    hipLaunchKernelGGL(k_mme7_selftest, dim3(1), dim3(64), 0, ctx->stream, d);
    (void) hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, ctx->stream);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipFree(d);
    result = h < 0 ? 1 : h;
    return result;
}

// Builds (or re-uses) the feature store of cloud `c` for radius grid + `radius`; *usable = false when the cloud does not fit the
// fixed-point frame (the caller then runs the vector kernel).
int mme7_prepare(me_ctx *ctx, Cloud &c, double radius, bool *usable) {
    *usable = false;
    if (c.mme_feat_valid && c.mme_feat_radius == radius) {
        *usable = c.mme_feat_usable;
        return ME_OK;
    }
    if (mme7_selftest(ctx) != 0)
        return ctx->fail(ME_ERR_HIP, "me_mme: the matrix-instruction self-test failed on this device (operand layout of "
                                     "v_mfma_i32_16x16x64_i8 / v_mfma_f32_16x16x4_f32, v_permlane*_swap or v_perm_b32 differs from gfx950's)");
    c.mme_feat_valid = false;
    const long long n_cells = c.grid.n_cells;
    TimerScope ts(ctx, "mme_feat");
    DevBuf &nch = ctx->tmp[0];
    ME_CHECK(ctx, nch.ensure((size_t) (n_cells + 2) * 4));
    ME_CHECK(ctx, c.mme_cell_chunk.ensure((size_t) (n_cells + 2) * 4));
    ME_CHECK(ctx, ctx->red.ensure(64));
    unsigned int *d_pmax = ctx->red.as<unsigned int>();
    ME_CHECK(ctx, hipMemsetAsync(d_pmax, 0, 4, ctx->stream));
    const unsigned int nbc = (unsigned int) ((n_cells + 1 + 255) / 256);
    hipLaunchKernelGGL(k_mme_chunks, dim3(nbc), dim3(256), 0, ctx->stream, c.grid.cell_start, n_cells, nch.as<unsigned int>(), d_pmax);
    ME_TRY(exclusive_scan_u32(ctx, nch.as<unsigned int>(), c.mme_cell_chunk.as<unsigned int>(), n_cells + 1));
    unsigned int h_total = 0, h_pmax = 0;
    ME_CHECK(ctx, hipMemcpyAsync(&h_total, c.mme_cell_chunk.as<unsigned int>() + n_cells, 4, hipMemcpyDeviceToHost, ctx->stream));
    ME_CHECK(ctx, hipMemcpyAsync(&h_pmax, d_pmax, 4, hipMemcpyDeviceToHost, ctx->stream));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // scale: every accepted neighbour of a query lies in its 27 cells
    double max_abs = 0, extent = 0;
    for (int d = 0; d < 3; ++d) {
        max_abs = std::fmax(max_abs, std::fmax(std::fabs(c.origin[d]), std::fmax(std::fabs(c.bbox_lo[d]), std::fabs(c.bbox_hi[d]))));
        extent = std::fmax(extent, c.bbox_hi[d] - c.origin[d]);
    }
    const double kmax = std::fmin((double) c.n, 27.0 * (double) h_pmax);
    const int s = fx::choose_scale(max_abs, extent, radius, kmax);
    c.mme_feat_radius = radius;
    c.mme_feat_valid = true;
    c.mme_feat_usable = false;
    // (int32 digit sums: |sum| <= 128 k;  chunk ids carry 27 bits in the list entries)
    if (s < 0 || kmax >= 0x1p23 || h_total >= (1u << 27)) return ME_OK;
    c.mme_fx_scale = s;
    for (int d = 0; d < 3; ++d) c.mme_fx_origin[d] = fx::fix(c.origin[d], s);
    const long long n_chunks = h_total;
    ME_CHECK(ctx, c.mme_feat.ensure((size_t) (n_chunks + 4) * fx::kChunkBytes));
    DevBuf &desc = ctx->tmp[1];
    ME_CHECK(ctx, desc.ensure((size_t) (n_chunks + 1) * 8));
    hipLaunchKernelGGL(k_mme_chunk_desc, dim3((unsigned int) ((n_cells + 255) / 256)), dim3(256), 0, ctx->stream, c.grid.cell_start,
                       c.mme_cell_chunk.as<unsigned int>(), n_cells, desc.as<uint2>());
    const fx::Frame f{c.mme_fx_origin[0], c.mme_fx_origin[1], c.mme_fx_origin[2], s};
    hipLaunchKernelGGL(k_mme_feat, dim3((unsigned int) ((n_chunks + 15) / 16)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(),
                       desc.as<uint2>(), n_chunks, f, c.mme_feat.as<unsigned int>());
    ME_CHECK(ctx, hipGetLastError());
    c.mme_feat_usable = true;
    *usable = true;
    return ME_OK;
}

int mme7_launch(me_ctx *ctx, Cloud &c, long long b, long long e, unsigned int nb, double radius, int min_k, double *ent_s,
                unsigned char *valid_s, double *part_sum, long long *part_cnt) {
    const FrameView fr{c.origin[0], c.origin[1], c.origin[2], c.fine_h};
    const fx::Frame f{c.mme_fx_origin[0], c.mme_fx_origin[1], c.mme_fx_origin[2], c.mme_fx_scale};
    const double r2 = radius * radius;  // Open3D SearchRadius -> nanoflann radiusSearch(q, r*r) [upstream]
    const float band = (float) (0x1p-12 * c.cell_h * c.cell_h);  // E
    const float thr_hi = (float) r2 + band;
    const float band2 = 2.0f * band;
    unsigned int band_bits;
    std::memcpy(&band_bits, &band2, 4);
    hipLaunchKernelGGL(k_mme7, dim3(nb), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), c.codes.as<unsigned long long>(), b, e, c.grid,
                       c.mme_cell_chunk.as<unsigned int>(), c.mme_feat.as<unsigned char>(), fr, f, c.slab, r2, min_k, ent_s, valid_s,
                       part_sum, part_cnt, xcd_chunk_setting(), c.cell_h, thr_hi, band_bits);
    return ME_OK;
}

}  // namespace me
#endif  // ME_AB
