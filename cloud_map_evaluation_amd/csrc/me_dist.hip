// me_dist.hip — device-side pieces of the multi-GPU step (no reference counterpart: the reference is one process).
//
//   halo_pack          the send side of the one-shot halo exchange: every point of this rank's part of a cloud is copied
//                      into the send segment of EVERY rank whose slab (+ halo) contains it — a deterministic multi-split,
//                      destination-major, ready for one all_to_all.
//   voxel_rows_device  this rank's voxel partials as rows on the device (what the all-gather carries).
//   voxel_merge        Chan's parallel update of the gathered partials -> the cloud's voxel table exactly as
//                      VoxelCalculator::buildVoxelMap leaves it (voxel_calculator.cpp:21-56), so that me_awd_scs runs on it.
#include <cmath>
#include <cstring>
#include <vector>

#include "me_internal.hpp"

namespace me {

constexpr int kSplitPerLane = 32;  // a wavefront splits 64 x 32 = 2048 consecutive points
constexpr int kSplitUnit = 64 * kSplitPerLane;
constexpr int kMaxWorld = 64;      // one lane of the wavefront keeps the running count of one destination

struct Cuts {
    double lo[kMaxWorld], hi[kMaxWorld];  // [lo_k, hi_k) = slab k grown by the halo: the filter me_set_slab applies
};

// does destination k hold coordinate v?  (the expression of k_slab_flags: v >= reg_lo && v < reg_hi)
__device__ __forceinline__ bool slab_holds(const Cuts &c, int k, double v) { return v >= c.lo[k] && v < c.hi[k]; }

template <bool SCATTER>
__global__ void __launch_bounds__(256)
k_halo_split(const double *__restrict__ xyz, long long n, int axis, Cuts cuts, int world, long long n_units,
             unsigned int *__restrict__ unit_counts /* [world][n_units] */, const unsigned int *__restrict__ unit_off,
             double *__restrict__ out, long long *__restrict__ tag_out, long long tag_base) {
    // tag_out (optional): tag_base + the point's input index travels with it — what the receiving rank needs to say WHICH point
    // of the whole cloud an entry of its per-point outputs is (the distributed host's map_entropy.pcd / raw_rendered_dis_map.pcd)
    const int lane = threadIdx.x & 63;
    const long long u = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_units) return;
    // lane k keeps the running count (SCATTER: the running output position) of destination k
    unsigned int mine = 0;
    if (SCATTER && lane < world) mine = unit_off[(long long) lane * n_units + u];
    for (int it = 0; it < kSplitPerLane; ++it) {
        const long long i = u * kSplitUnit + (long long) it * 64 + lane;
        const bool live = i < n;
        double p[3] = {0, 0, 0};
        if (live) {
            p[0] = xyz[3 * i];
            p[1] = xyz[3 * i + 1];
            p[2] = xyz[3 * i + 2];
        }
        const double v = p[axis];
        for (int k = 0; k < world; ++k) {
            const bool member = live && slab_holds(cuts, k, v);
            const unsigned long long m = __ballot(member);
            if (!m) continue;
            if (SCATTER) {
                const unsigned int base = (unsigned int) __builtin_amdgcn_readlane((int) mine, k);
                if (member) {
                    const long long o = (long long) base + __popcll(m & ((1ULL << lane) - 1ULL));
                    out[3 * o] = p[0];
                    out[3 * o + 1] = p[1];
                    out[3 * o + 2] = p[2];
                    if (tag_out) tag_out[o] = tag_base + i;
                }
            }
            if (lane == k) mine += (unsigned int) __popcll(m);
        }
    }
    if (!SCATTER && lane < world) unit_counts[(long long) lane * n_units + u] = mine;
}

__global__ void k_split_segments(const unsigned int *__restrict__ off, const unsigned int *__restrict__ cnt, long long n_units,
                                 int world, unsigned int *__restrict__ seg) {
    const int k = threadIdx.x;
    if (k < world) seg[k] = off[(long long) k * n_units];
    if (k == world) seg[world] = off[n_units * world - 1] + cnt[n_units * world - 1];
}

// 64-bit totals per destination (the segment offsets are 32-bit: a point is copied to every rank whose slab + halo holds it,
// so the packed total can exceed 2^32 although n does not; the host checks these before it trusts the offsets)
__global__ void __launch_bounds__(256)
k_split_totals(const unsigned int *__restrict__ cnt, long long n_units, unsigned long long *__restrict__ totals) {
    const int k = blockIdx.x;
    unsigned long long s = 0;
    for (long long u = threadIdx.x; u < n_units; u += 256) s += cnt[(long long) k * n_units + u];
    __shared__ long long sm[4];
    const long long t = block_sum_256_ll((long long) s, sm);
    if (threadIdx.x == 0) totals[k] = (unsigned long long) t;
}

int halo_pack(me_ctx *ctx, const double *xyz_device, long long n, int axis, const double *cuts_host, int world, double halo,
              double *out_device, long long capacity, long long *counts_host, long long *tags_device, long long tag_base) {
    if ((n > 0 && !xyz_device) || n < 0 || axis < 0 || axis > 2 || !cuts_host || world < 1 || world > kMaxWorld || !(halo >= 0) || !counts_host)
        return ctx->fail(ME_ERR_ARG, "me_halo_pack_device: bad argument (1 <= world <= 64)");
    if (n >= (1LL << 31)) return ctx->fail(ME_ERR_ARG, "me_halo_pack_device: more than 2^31 - 1 points in one call");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    Cuts c{};
    for (int k = 0; k < world; ++k) {
        if (!(cuts_host[k] < cuts_host[k + 1])) return ctx->fail(ME_ERR_ARG, "me_halo_pack_device: cuts must be strictly ascending");
        c.lo[k] = cuts_host[k] - halo;      // me_set_slab: reg_lo = lo - halo
        c.hi[k] = cuts_host[k + 1] + halo;  //              reg_hi = hi + halo
    }
    for (int k = 0; k < world; ++k) counts_host[k] = 0;
    if (n == 0) return ME_OK;
    const long long n_units = (n + kSplitUnit - 1) / kSplitUnit;
    const long long n_cnt = n_units * world;
    DevBuf &cnt = ctx->tmp[0], &off = ctx->tmp[1];
    ME_CHECK(ctx, cnt.ensure((size_t) n_cnt * 4));
    ME_CHECK(ctx, off.ensure((size_t) n_cnt * 4));
    const dim3 grid((unsigned int) ((n_units + 3) / 4));
    TimerScope ts(ctx, "halo_pack");
    hipLaunchKernelGGL(k_halo_split<false>, grid, dim3(256), 0, ctx->stream, xyz_device, n, axis, c, world, n_units,
                       cnt.as<unsigned int>(), (const unsigned int *) nullptr, (double *) nullptr, (long long *) nullptr, 0LL);
    // the exact 64-bit totals first: 32-bit offsets are only meaningful when the packed total fits them
    ME_CHECK(ctx, ctx->tmp[2].ensure((size_t) world * 8));
    hipLaunchKernelGGL(k_split_totals, dim3((unsigned int) world), dim3(256), 0, ctx->stream, cnt.as<unsigned int>(), n_units,
                       ctx->tmp[2].as<unsigned long long>());
    std::vector<unsigned long long> tot64((size_t) world);
    ME_TRY(copy_d2h(ctx, tot64.data(), ctx->tmp[2].p, (size_t) world * 8));
    ME_TRY(exclusive_scan_u32(ctx, cnt.as<unsigned int>(), off.as<unsigned int>(), n_cnt));
    // destination k's segment starts at off[k * n_units]; the total is off[last] + cnt[last]: one small kernel collects the
    // world + 1 numbers, ONE copy brings them to the host (world separate 4-byte copies cost ~10 us each)
    ME_CHECK(ctx, ctx->red.ensure((size_t) (world + 1) * 4));
    hipLaunchKernelGGL(k_split_segments, dim3(1), dim3(kMaxWorld + 1), 0, ctx->stream, off.as<unsigned int>(), cnt.as<unsigned int>(),
                       n_units, world, ctx->red.as<unsigned int>());
    std::vector<unsigned int> seg((size_t) world + 1);
    ME_TRY(copy_d2h(ctx, seg.data(), ctx->red.p, (size_t) (world + 1) * 4));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    long long total = 0;
    for (int k = 0; k < world; ++k) {
        counts_host[k] = (long long) tot64[(size_t) k];
        total += counts_host[k];
    }
    if (total > 0xffffffffLL) {
        if (!out_device) return ME_OK;  // counts only: they are exact; packing needs 32-bit offsets
        return ctx->fail(ME_ERR_CAPACITY, "me_halo_pack_device: the packed total exceeds 2^32 - 1 points (split the call)");
    }
    for (int k = 0; k < world; ++k)
        if (counts_host[k] != (long long) seg[k + 1] - (long long) seg[k])
            return ctx->fail(ME_ERR_STATE, "me_halo_pack_device: segment offsets disagree with the 64-bit totals");
    if (!out_device) return ME_OK;  // counts only
    if (capacity < total) return ctx->fail(ME_ERR_CAPACITY, "me_halo_pack_device: capacity too small (counts returned)");
    hipLaunchKernelGGL(k_halo_split<true>, grid, dim3(256), 0, ctx->stream, xyz_device, n, axis, c, world, n_units,
                       (unsigned int *) nullptr, off.as<unsigned int>(), out_device, tags_device, tag_base);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

// ---- marginal histograms on an absolute power-of-two lattice (the lean exchange of dist.py) ----
// A rank that knows, for every rank's part of a cloud, how many points fall into every bin [i w, (i + 1) w) of every axis can compute
// the slab cuts (at bin edges), the halo (a whole number of bins) AND the exact size of every message of the halo exchange by itself:
// one all-gather of these histograms replaces the sample gather, the count all-to-all and three host reads.  w is a power of two and
// the lattice is absolute (bin = floor(x / w)), so that k_halo_split's comparisons v >= lo, v < hi against lo, hi = (whole number) * w
// decide exactly what the bins say: x / w, floor and (whole number) * w are all exact in fp64.
constexpr long long kLatClamp = 1LL << 60;

__device__ __forceinline__ long long lattice_bin(double v, double inv_w) {
    const double f = floor(v * inv_w);
    return f >= (double) kLatClamp ? kLatClamp : (f <= -(double) kLatClamp ? -kLatClamp : (long long) f);
}

// range[0..2] = min, range[3..5] = max of the level-0 bin over the FINITE coordinates of every axis; range[6..8] = how many are -inf
// (k_halo_split gives those to rank 0: -inf >= -inf; NaN and +inf satisfy no slab's test and are dropped by the exchange)
__global__ void __launch_bounds__(256)
k_lattice_range(const double *__restrict__ xyz, long long n, double inv_w, long long *__restrict__ range) {
    long long mn[3] = {kLatClamp, kLatClamp, kLatClamp}, mx[3] = {-kLatClamp, -kLatClamp, -kLatClamp}, ni[3] = {0, 0, 0};
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const double v = xyz[3 * i + a];
            if (isfinite(v)) {
                const long long b = lattice_bin(v, inv_w);
                mn[a] = min(mn[a], b);
                mx[a] = max(mx[a], b);
            } else if (v < 0) {
                ++ni[a];
            }
        }
    __shared__ long long sm[9][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a) {
        long long lo = mn[a], hi = mx[a], c = ni[a];
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, (long long) __shfl_xor(lo, o, 64));
            hi = max(hi, (long long) __shfl_xor(hi, o, 64));
            c += (long long) __shfl_xor(c, o, 64);
        }
        if (lane == 0) {
            sm[a][wv] = lo;
            sm[3 + a][wv] = hi;
            sm[6 + a][wv] = c;
        }
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int r = threadIdx.x;
        long long v = sm[r][0];
        for (int w = 1; w < 4; ++w) v = r < 3 ? min(v, sm[r][w]) : (r < 6 ? max(v, sm[r][w]) : v + sm[r][w]);
        if (r < 3) atomicMin(&range[r], v);
        else if (r < 6) atomicMax(&range[r], v);
        else if (v) atomicAdd(reinterpret_cast<unsigned long long *>(&range[r]), (unsigned long long) v);
    }
}

struct LatOrigin {
    long long o[3];
};

// hist[a * bins + (floor(x_a / w) - origin_a)] += 1 for every finite coordinate (the caller chose w and the origin so that all fit);
// block-private counts in LDS, one global atomic per touched bin and block
template <int BINS>
__global__ void __launch_bounds__(256)
k_lattice_hist(const double *__restrict__ xyz, long long n, double inv_w, LatOrigin org, unsigned int *__restrict__ hist,
               unsigned int *__restrict__ outside) {
    __shared__ unsigned int sh[3 * BINS];
    for (int i = threadIdx.x; i < 3 * BINS; i += 256) sh[i] = 0;
    __syncthreads();
    unsigned int out = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const double v = xyz[3 * i + a];
            if (!isfinite(v)) continue;
            const long long b = lattice_bin(v, inv_w) - org.o[a];
            if (b >= 0 && b < BINS) atomicAdd(&sh[a * BINS + (int) b], 1u);
            else ++out;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * BINS; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
    if (out) atomicAdd(outside, out);
}

int lattice_histograms(me_ctx *ctx, const double *xyz_device, long long n, int e0, int *level, long long origin_bin[3], long long neg_inf[3],
                       unsigned int *hist_device) {
    if ((n > 0 && !xyz_device) || n < 0 || e0 < -40 || e0 > 40 || !level || !origin_bin || !neg_inf || !hist_device)
        return ctx->fail(ME_ERR_ARG, "me_lattice_histograms_device: bad argument (-40 <= e0 <= 40)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    constexpr int B = ME_LATTICE_BINS;
    *level = 0;
    for (int a = 0; a < 3; ++a) origin_bin[a] = neg_inf[a] = 0;
    ME_CHECK(ctx, hipMemsetAsync(hist_device, 0, (size_t) 3 * B * 4, ctx->stream));
    if (n == 0) {
        ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return ME_OK;
    }
    TimerScope ts(ctx, "halo_pack");
    ME_CHECK(ctx, ctx->red.ensure(16 * 8));
    long long *d_range = ctx->red.as<long long>();
    long long init[10] = {kLatClamp, kLatClamp, kLatClamp, -kLatClamp, -kLatClamp, -kLatClamp, 0, 0, 0, 0};
    ME_CHECK(ctx, hipMemcpyAsync(d_range, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    const unsigned int grid = (unsigned int) std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_lattice_range, dim3(grid), dim3(256), 0, ctx->stream, xyz_device, n, std::ldexp(1.0, -e0), d_range);
    long long h[9];
    {
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, h, d_range, sizeof h));
        ME_TRY(mg.sync());
    }
    for (int a = 0; a < 3; ++a) neg_inf[a] = h[6 + a];
    bool any = false;
    for (int a = 0; a < 3; ++a) any = any || h[a] <= h[3 + a];
    if (any) {
        int L = 0;
        auto fits = [&](int l) {
            for (int a = 0; a < 3; ++a)
                if (h[a] <= h[3 + a] && (h[3 + a] >> l) - (h[a] >> l) + 1 > (long long) B) return false;
            return true;
        };
        while (L < 62 && !fits(L)) ++L;
        *level = L;
        LatOrigin org{};
        for (int a = 0; a < 3; ++a) org.o[a] = origin_bin[a] = (h[a] <= h[3 + a]) ? (h[a] >> L) : 0;
        unsigned int *d_out = reinterpret_cast<unsigned int *>(d_range + 9);
        hipLaunchKernelGGL((k_lattice_hist<B>), dim3(std::min(grid, 256u)), dim3(256), 0, ctx->stream, xyz_device, n, std::ldexp(1.0, -(e0 + L)), org,
                           hist_device, d_out);
        unsigned int outside = 0;
        {
            MailGuard mg(ctx);
            ME_TRY(mail_post(ctx, &outside, d_out, 4));
            ME_TRY(mg.sync());
        }
        if (outside) return ctx->fail(ME_ERR_STATE, "me_lattice_histograms_device: points outside the window the range pass chose");
    }
    ME_CHECK(ctx, hipGetLastError());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

// ... and the two pieces of a rank at once, as the rows of its gather message (header included): two launches per piece, ONE host read
// (both ranges) and one stream synchronisation for both — what dist.lattice_message costs at the head of every step
template <int BINS>
__global__ void __launch_bounds__(1024)
k_lattice_row(const double *__restrict__ xyz, long long n, double inv_w, LatOrigin org, long long level, const long long *__restrict__ range,
              long long *__restrict__ row /* 8 + 3 BINS */) {
    __shared__ unsigned int sh[3 * BINS];
    for (int i = threadIdx.x; i < 3 * BINS; i += (int) blockDim.x) sh[i] = 0;
    __syncthreads();
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const double v = xyz[3 * i + a];
            if (!isfinite(v)) continue;
            const long long b = lattice_bin(v, inv_w) - org.o[a];
            if (b >= 0 && b < BINS) atomicAdd(&sh[a * BINS + (int) b], 1u);  // (always: the window was chosen from the range of these points)
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * BINS; i += (int) blockDim.x)
        if (sh[i]) atomicAdd(reinterpret_cast<unsigned long long *>(row + 8 + i), (unsigned long long) sh[i]);
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        const int t = threadIdx.x;
        row[t] = t == 0 ? level : (t < 4 ? org.o[t - 1] : (t == 4 ? n : range[6 + (t - 5)]));
    }
}

int lattice_messages(me_ctx *ctx, const double *const xyz_device[2], const long long n[2], int clouds, int e0, long long *msg_device) {
    if (clouds < 1 || clouds > 2 || !xyz_device || !n || e0 < -40 || e0 > 40 || !msg_device)
        return ctx->fail(ME_ERR_ARG, "me_lattice_messages_device: bad argument (1 <= clouds <= 2, -40 <= e0 <= 40)");
    for (int c = 0; c < clouds; ++c)
        if (n[c] < 0 || (n[c] > 0 && !xyz_device[c])) return ctx->fail(ME_ERR_ARG, "me_lattice_messages_device: bad cloud");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    constexpr int B = ME_LATTICE_BINS;
    constexpr size_t row_len = 8 + 3 * (size_t) B;
    TimerScope ts(ctx, "halo_pack");
    ME_CHECK(ctx, hipMemsetAsync(msg_device, 0, (size_t) clouds * row_len * 8, ctx->stream));
    ME_CHECK(ctx, ctx->red.ensure(2 * 10 * 8));
    long long *d_range = ctx->red.as<long long>();
    long long init[20];
    for (int c = 0; c < 2; ++c)
        for (int k = 0; k < 10; ++k) init[c * 10 + k] = k < 3 ? kLatClamp : (k < 6 ? -kLatClamp : 0);
    ME_CHECK(ctx, hipMemcpyAsync(d_range, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    for (int c = 0; c < clouds; ++c)
        if (n[c] > 0)
            hipLaunchKernelGGL(k_lattice_range, dim3((unsigned int) std::min<long long>((n[c] + 255) / 256, 1024)), dim3(256), 0, ctx->stream,
                               xyz_device[c], n[c], std::ldexp(1.0, -e0), d_range + c * 10);
    long long h[20];
    {
        MailGuard mg(ctx);
        ME_TRY(mail_post(ctx, h, d_range, sizeof h));
        ME_TRY(mg.sync());
    }
    for (int c = 0; c < clouds; ++c) {
        const long long *hc = h + c * 10;
        int L = 0;
        auto fits = [&](int l) {
            for (int a = 0; a < 3; ++a)
                if (hc[a] <= hc[3 + a] && (hc[3 + a] >> l) - (hc[a] >> l) + 1 > (long long) B) return false;
            return true;
        };
        while (L < 62 && !fits(L)) ++L;
        LatOrigin org{};
        for (int a = 0; a < 3; ++a) org.o[a] = (hc[a] <= hc[3 + a]) ? (hc[a] >> L) : 0;
        // (an empty piece: one block writes the header — level 0, origin 0, n = 0 — and no counts)
        // (one 48 KB histogram per CU, sixteen wavefronts reading for it: with four the pass ran at 1.3 TB/s)
        const unsigned int grid = n[c] > 0 ? (unsigned int) std::min<long long>((n[c] + 1023) / 1024, 256) : 1u;
        hipLaunchKernelGGL((k_lattice_row<B>), dim3(grid), dim3(1024), 0, ctx->stream, xyz_device[c], n[c], std::ldexp(1.0, -(e0 + L)), org,
                           (long long) L, (const long long *) (d_range + c * 10), msg_device + (size_t) c * row_len);
    }
    ME_CHECK(ctx, hipGetLastError());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

// ---- the plan of the lean exchange from the gathered histogram messages (dist.lattice_plan restated on the device) ----
// msgs: world x clouds rows of [level, origin_x, origin_y, origin_z, n, neg_inf_x, neg_inf_y, neg_inf_z | 3 x BINS counts] (int64).
// Every step mirrors dist.lattice_plan (the torch form the CPU tests run) number for number.
constexpr int kPlanG = 2 * ME_LATTICE_BINS;  // bins of the combined window
constexpr int kPlanRow = 8 + 3 * ME_LATTICE_BINS;
struct PlanHead {  // written by k_plan_extents, read by the two kernels after it
    long long K, e, g0[3], axis, m, any_ref;
};

// first / last occupied bin of one (row, axis) per block
__global__ void __launch_bounds__(256)
k_plan_rowext(const long long *__restrict__ msgs, long long *__restrict__ rowext /* [rows][3][2] */) {
    constexpr int B = ME_LATTICE_BINS;
    const int ra = blockIdx.x;
    const long long *h = msgs + (long long) (ra / 3) * kPlanRow + 8 + (long long) (ra % 3) * B;
    int lo = B, hi = -1;
    for (int i = threadIdx.x; i < B; i += 256)
        if (h[i] > 0) {
            lo = min(lo, i);
            hi = max(hi, i);
        }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    __shared__ int s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        rowext[ra * 2] = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        rowext[ra * 2 + 1] = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
    }
}

__global__ void k_plan_extents(const long long *__restrict__ msgs, int world, int clouds, double halo, int e0,
                               const long long *__restrict__ rowext, PlanHead *__restrict__ head) {
    const int rows = world * clouds;
    if (threadIdx.x != 0) return;
    long long K = 0;
    for (int r = 0; r < rows; ++r) K = max(K, msgs[(long long) r * kPlanRow]);
    const long long big = 1LL << 61;
    long long amin[3], amax[3], gmin[3], gmax[3];  // occupied stretch per axis at level K: all rows / the ground truth's rows
    bool any_gt = false;
    for (int a = 0; a < 3; ++a) {
        amin[a] = gmin[a] = big;
        amax[a] = gmax[a] = -big;
    }
    for (int r = 0; r < rows; ++r) {
        const long long *mr = msgs + (long long) r * kPlanRow;
        const long long sh = K - mr[0];
        for (int a = 0; a < 3; ++a) {
            const long long lo = rowext[(r * 3 + a) * 2], hi = rowext[(r * 3 + a) * 2 + 1];
            if (hi < 0) continue;
            const long long alo = (mr[1 + a] + lo) >> sh, ahi = (mr[1 + a] + hi) >> sh;
            amin[a] = min(amin[a], alo);
            amax[a] = max(amax[a], ahi);
            if (r % clouds == clouds - 1) {  // the ground truth's pieces
                gmin[a] = min(gmin[a], alo);
                gmax[a] = max(gmax[a], ahi);
                any_gt = true;
            }
        }
    }
    // the slab axis: the longest occupied stretch of the ground truth (of everything when no rank holds any of it); an axis without an
    // occupied bin in that set: -1 - G, as dist.lattice_plan's masked max - min gives
    long long rlo[3], rhi[3];
    for (int a = 0; a < 3; ++a) {
        rlo[a] = any_gt ? gmin[a] : amin[a];
        rhi[a] = any_gt ? gmax[a] : amax[a];
    }
    long long span = 1;
    for (int a = 0; a < 3; ++a) {
        if (amin[a] > amax[a]) amin[a] = amax[a] = 0;  // (an axis nobody has a finite coordinate on)
        span = max(span, amax[a] - amin[a] + 1);
    }
    long long e = 0;
    while (e < 50 && ((long long) (kPlanG - 1) << e) < span) ++e;  // (2^13 << 49 < 2^63; bins are clamped to +-2^60)
    for (int it = 0; it < 2; ++it) {
        long long se = 0;
        for (int a = 0; a < 3; ++a) se = max(se, (amax[a] >> e) - (amin[a] >> e) + 1);
        if (se > kPlanG) ++e;
    }
    long long best = 0, best_ext = 0;
    for (int a = 0; a < 3; ++a) {
        const long long ext = rlo[a] <= rhi[a] ? (rhi[a] >> e) - (rlo[a] >> e) : (long long) (-1 - kPlanG);
        if (a == 0 || ext > best_ext) {
            best = a;
            best_ext = ext;
        }
    }
    head->K = K;
    head->e = e;
    for (int a = 0; a < 3; ++a) head->g0[a] = amin[a] >> e;
    head->axis = best;
    const double mm = ceil(halo * ldexp(1.0, -(int) (K + e + e0)));
    head->m = mm < 1.0 ? 1 : (long long) mm;
    head->any_ref = any_gt ? 1 : 0;
}

// one block per (rank, cloud): its histogram of the slab axis re-binned into the combined window -> exclusive prefix P[0 .. G]
__global__ void __launch_bounds__(256)
k_plan_prefix(const long long *__restrict__ msgs, const PlanHead *__restrict__ head, long long *__restrict__ P /* [rows][G + 1] */) {
    constexpr int B = ME_LATTICE_BINS, G = kPlanG;
    __shared__ unsigned long long sh[G];
    __shared__ unsigned long long part[256];
    const int r = blockIdx.x;
    const long long *mr = msgs + (long long) r * kPlanRow;
    const int axis = (int) head->axis;
    const long long shf = head->K - mr[0], e = head->e, g0 = head->g0[axis], org = mr[1 + axis];
    const long long *h = mr + 8 + (long long) axis * B;
    for (int i = threadIdx.x; i < G; i += 256) sh[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const long long v = h[i];
        if (v > 0) {
            long long idx = (((org + i) >> shf) >> e) - g0;
            idx = idx < 0 ? 0 : (idx > G - 1 ? G - 1 : idx);
            atomicAdd(&sh[idx], (unsigned long long) v);
        }
    }
    __syncthreads();
    constexpr int PER = G / 256;
    unsigned long long run = 0;
    for (int j = 0; j < PER; ++j) run += sh[threadIdx.x * PER + j];
    part[threadIdx.x] = run;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int t = 0; t < 256; ++t) {
            const unsigned long long v = part[t];
            part[t] = acc;
            acc += v;
        }
    }
    __syncthreads();
    unsigned long long acc = part[threadIdx.x];
    long long *Pr = P + (long long) r * (G + 1);
    for (int j = 0; j < PER; ++j) {
        Pr[threadIdx.x * PER + j] = (long long) acc;
        acc += sh[threadIdx.x * PER + j];
    }
    if (threadIdx.x == 255) Pr[G] = (long long) acc;
}

// cuts at the k / world quantiles of all rows' counts, then every (source row, destination) count
__global__ void __launch_bounds__(256)
k_plan_cuts_counts(const long long *__restrict__ msgs, const PlanHead *__restrict__ head, const long long *__restrict__ P, int world, int clouds,
                   long long *__restrict__ out) {
    constexpr int G = kPlanG;
    const int rows = world * clouds;
    __shared__ long long s_c[kMaxWorld + 1], s_lo[kMaxWorld], s_hi[kMaxWorld];
    auto cum = [&](int j) {  // points of all rows in bins [0, j]
        long long t = 0;
        for (int r = 0; r < rows; ++r) t += P[(long long) r * (G + 1) + j + 1];
        return t;
    };
    const long long N = cum(G - 1);
    if ((int) threadIdx.x < world - 1) {
        const long long k = threadIdx.x + 1;
        long long target = (N * k + world - 1) / world;
        if (target < 1) target = 1;
        int lo = 0, hi = G;  // first j with cum(j) >= target (G: none)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum(mid) >= target) hi = mid;
            else lo = mid + 1;
        }
        s_c[k] = lo + 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = -(1LL << 60);
        for (int k = 1; k < world; ++k) {  // strictly ascending: cummax(c - step) + step
            run = max(run, s_c[k] - (k - 1));
            s_c[k] = run + (k - 1);
        }
        const long long m = head->m;
        for (int k = 0; k < world; ++k) {
            long long lo = k == 0 ? 0 : s_c[k] - m, hi = k == world - 1 ? G : s_c[k + 1] + m;
            s_lo[k] = lo < 0 ? 0 : (lo > G ? G : lo);
            s_hi[k] = hi < 0 ? 0 : (hi > G ? G : hi);
        }
        out[0] = head->axis;
        out[1] = head->K + head->e;
        out[2] = head->g0[head->axis];
        out[3] = m;
        for (int c = 0; c < clouds; ++c) {
            long long t = 0;
            for (int w = 0; w < world; ++w) t += msgs[(long long) (w * clouds + c) * kPlanRow + 4];
            out[4 + c] = t;
        }
        for (int k = 1; k < world; ++k) out[4 + clouds + k - 1] = s_c[k];
    }
    __syncthreads();
    const int axis = (int) head->axis;
    for (int t = threadIdx.x; t < rows * world; t += 256) {
        const int r = t / world, k = t % world;
        long long v = P[(long long) r * (G + 1) + s_hi[k]] - P[(long long) r * (G + 1) + s_lo[k]];
        if (v < 0) v = 0;
        if (k == 0) v += msgs[(long long) r * kPlanRow + 5 + axis];
        out[4 + clouds + world - 1 + t] = v;
    }
}

int lattice_plan(me_ctx *ctx, const long long *msgs_device, int world, int clouds, double halo, int e0, long long *out_host) {
    if (!msgs_device || world < 1 || world > kMaxWorld || clouds < 1 || clouds > 2 || !(halo > 0) || e0 < -40 || e0 > 40 || !out_host)
        return ctx->fail(ME_ERR_ARG, "me_lattice_plan_device: bad argument (1 <= world <= 64, 1 <= clouds <= 2, halo > 0)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const int rows = world * clouds;
    const size_t n_out = (size_t) 4 + clouds + (world - 1) + (size_t) rows * world;
    DevBuf &ext = ctx->tmp[0], &pre = ctx->tmp[1], &outb = ctx->tmp[2];
    ME_CHECK(ctx, ext.ensure((size_t) rows * 6 * 8 + sizeof(PlanHead)));
    ME_CHECK(ctx, pre.ensure((size_t) rows * (kPlanG + 1) * 8));
    ME_CHECK(ctx, outb.ensure(n_out * 8));
    long long *d_ext = ext.as<long long>();
    PlanHead *d_head = reinterpret_cast<PlanHead *>(d_ext + (size_t) rows * 6);
    TimerScope ts(ctx, "halo_pack");
    hipLaunchKernelGGL(k_plan_rowext, dim3((unsigned int) rows * 3), dim3(256), 0, ctx->stream, msgs_device, d_ext);
    hipLaunchKernelGGL(k_plan_extents, dim3(1), dim3(1), 0, ctx->stream, msgs_device, world, clouds, halo, e0, (const long long *) d_ext, d_head);
    hipLaunchKernelGGL(k_plan_prefix, dim3((unsigned int) rows), dim3(256), 0, ctx->stream, msgs_device, (const PlanHead *) d_head, pre.as<long long>());
    hipLaunchKernelGGL(k_plan_cuts_counts, dim3(1), dim3(256), 0, ctx->stream, msgs_device, (const PlanHead *) d_head, (const long long *) pre.as<long long>(),
                       world, clouds, outb.as<long long>());
    ME_CHECK(ctx, hipGetLastError());
    ME_TRY(copy_d2h(ctx, out_host, outb.p, n_out * 8));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

// ---- voxel partial rows: [kx, ky, kz, n, mu(3), M2(9)] = 16 doubles per voxel ----
constexpr int kRow = 16;

__device__ __forceinline__ void unpack3(unsigned long long k, int &kx, int &ky, int &kz) {
    const int bias = 1 << 20;
    kx = (int) ((k >> 42) & 0x1fffff) - bias;
    ky = (int) ((k >> 21) & 0x1fffff) - bias;
    kz = (int) (k & 0x1fffff) - bias;
}

__global__ void k_vox_rows(const unsigned long long *__restrict__ key, const int *__restrict__ vn, const double *__restrict__ mu,
                           const double *__restrict__ m2, long long V, double *__restrict__ rows) {
    const long long v = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int kx, ky, kz;
    unpack3(key[v], kx, ky, kz);
    double *r = rows + kRow * v;
    r[0] = kx;
    r[1] = ky;
    r[2] = kz;
    r[3] = vn[v];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[4 + k] = mu[3 * v + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[7 + k] = m2[9 * v + k];
}

int voxel_rows_device(me_ctx *ctx, int slot, double voxel_size, double *rows_device, long long capacity, long long *n_rows) {
    if (slot < 0 || slot > 1 || !n_rows) return ctx->fail(ME_ERR_ARG, "me_voxel_partial_rows_device: bad argument");
    ME_TRY(voxel_build(ctx, slot, voxel_size, true));
    Cloud &c = ctx->cloud[slot];
    *n_rows = c.n_vox;
    if (!rows_device || c.n_vox == 0) return ME_OK;
    if (capacity < c.n_vox) return ctx->fail(ME_ERR_CAPACITY, "me_voxel_partial_rows_device: capacity too small");
    hipLaunchKernelGGL(k_vox_rows, dim3((unsigned int) ((c.n_vox + 255) / 256)), dim3(256), 0, ctx->stream,
                       c.vox_key.as<unsigned long long>(), c.vox_n.as<int>(), c.vox_mu.as<double>(), c.vox_sigma.as<double>(),
                       c.n_vox, rows_device);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

// ---- merge of gathered partial rows ----
constexpr unsigned long long kRowSentinel = 0x7fffffffffffffffULL;  // padding rows (n == 0) sort last

__global__ void k_row_keys(const double *__restrict__ rows, long long m, unsigned long long *__restrict__ keys,
                           unsigned int *__restrict__ iota) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double *r = rows + kRow * i;
    const int bias = 1 << 20;
    unsigned long long k = kRowSentinel;
    if (r[3] > 0.0)
        k = ((unsigned long long) (unsigned int) ((int) r[0] + bias) << 42) | ((unsigned long long) (unsigned int) ((int) r[1] + bias) << 21) |
            (unsigned long long) (unsigned int) ((int) r[2] + bias);
    keys[i] = k;
    iota[i] = (unsigned int) i;
}

__global__ void k_row_heads(const unsigned long long *__restrict__ keys, long long m, unsigned int *__restrict__ flags) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_row_segments(const unsigned long long *__restrict__ keys, const unsigned int *__restrict__ flags,
                               const unsigned int *__restrict__ pos, long long m, unsigned long long *__restrict__ vkey,
                               unsigned int *__restrict__ seg_start) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (flags[i]) {
        vkey[pos[i]] = keys[i];
        seg_start[pos[i]] = (unsigned int) i;
    }
}

// One thread per voxel: n = sum n_i, mu = sum n_i mu_i / n, M2 = sum (M2_i + n_i (mu_i - mu)(mu_i - mu)^T) over its partials in
// (rank, row) order, then the reference's finalisation (two divisions for n > 10 and the entropy, voxel_calculator.cpp:48,102-109).
__global__ void k_vox_merge(const double *__restrict__ rows, const unsigned int *__restrict__ perm,
                            const unsigned int *__restrict__ seg_start, long long V, long long m_end, int *__restrict__ vn,
                            double *__restrict__ vmu, double *__restrict__ vsig, double *__restrict__ vent) {
    const long long v = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const long long b = seg_start[v], e = (v + 1 < V) ? (long long) seg_start[v + 1] : m_end;
    double n = 0, sx = 0, sy = 0, sz = 0;
    for (long long j = b; j < e; ++j) {
        const double *r = rows + kRow * (long long) perm[j];
        n += r[3];
        sx += r[3] * r[4];
        sy += r[3] * r[5];
        sz += r[3] * r[6];
    }
    const double mx = sx / n, my = sy / n, mz = sz / n;
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long long j = b; j < e; ++j) {
        const double *r = rows + kRow * (long long) perm[j];
        const double d[3] = {r[4] - mx, r[5] - my, r[6] - mz};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) S[3 * a + c] += r[7 + 3 * a + c] + r[3] * (d[a] * d[c]);
    }
    const long long cnt = (long long) n;
    double ent = 0.0;
    if (cnt > 10) {  // (:47)
        const double nm1 = (double) (cnt - 1);
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] = S[k] / nm1;  // (:48)
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] = S[k] / nm1;  // (:102)
        const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
        if (det > 0) {
            const double PI = 3.141592653589793238463;
            ent = 0.5 * log(pow(2 * PI * exp(1.0), 3.0) * det);  // (:109)
        }
    }
    vn[v] = (int) cnt;
    vmu[3 * v] = mx;
    vmu[3 * v + 1] = my;
    vmu[3 * v + 2] = mz;
#pragma unroll
    for (int k = 0; k < 9; ++k) vsig[9 * v + k] = S[k];
    vent[v] = ent;
}

int voxel_merge(me_ctx *ctx, int slot, double voxel_size, const double *rows_device, long long m) {
    if (slot < 0 || slot > 1 || m < 0 || (m > 0 && !rows_device) || !(voxel_size > 0))
        return ctx->fail(ME_ERR_ARG, "me_voxel_merge_device: bad argument");
    Cloud &c = ctx->cloud[slot];
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    c.vox_valid = false;
    c.vox_merged = false;
    long long V = 0;
    if (m > 0) {
        DevBuf &kin = ctx->tmp[0], &kout = ctx->tmp[1], &iota = ctx->tmp[2], &perm = ctx->tmp[3], &fl = ctx->tmp[4];
        ME_CHECK(ctx, kin.ensure((size_t) m * 8));
        ME_CHECK(ctx, kout.ensure((size_t) m * 8));
        ME_CHECK(ctx, iota.ensure((size_t) m * 4));
        ME_CHECK(ctx, perm.ensure((size_t) m * 4));
        ME_CHECK(ctx, fl.ensure((size_t) m * 8));
        unsigned int *flags = fl.as<unsigned int>(), *pos = flags + m;
        const dim3 g((unsigned int) ((m + 255) / 256));
        hipLaunchKernelGGL(k_row_keys, g, dim3(256), 0, ctx->stream, rows_device, m, kin.as<unsigned long long>(), iota.as<unsigned int>());
        // stable: the partials of one voxel stay in (rank, row) order -> a fixed summation order
        ME_TRY(sort_pairs_u64_u32(ctx, kin.as<unsigned long long>(), kout.as<unsigned long long>(), iota.as<unsigned int>(),
                                  perm.as<unsigned int>(), m, 0, 63));
        hipLaunchKernelGGL(k_row_heads, g, dim3(256), 0, ctx->stream, kout.as<unsigned long long>(), m, flags);
        ME_TRY(exclusive_scan_u32(ctx, flags, pos, m));
        unsigned int last_pos = 0, last_flag = 0;
        unsigned long long last_key = 0;
        ME_TRY(copy_d2h(ctx, &last_pos, pos + (m - 1), 4));
        ME_TRY(copy_d2h(ctx, &last_flag, flags + (m - 1), 4));
        ME_TRY(copy_d2h(ctx, &last_key, kout.as<unsigned long long>() + (m - 1), 8));
        ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const long long segs = (long long) last_pos + last_flag;
        const bool pad = last_key == kRowSentinel;
        V = pad ? segs - 1 : segs;
        ME_CHECK(ctx, c.vox_key.ensure((size_t) (segs + 1) * 8));
        ME_CHECK(ctx, c.vox_tmp.ensure((size_t) (segs + 2) * 4));
        ME_CHECK(ctx, c.vox_n.ensure((size_t) std::max<long long>(V, 1) * 4));
        ME_CHECK(ctx, c.vox_mu.ensure((size_t) std::max<long long>(V, 1) * 24));
        ME_CHECK(ctx, c.vox_sigma.ensure((size_t) std::max<long long>(V, 1) * 72));
        ME_CHECK(ctx, c.vox_entropy.ensure((size_t) std::max<long long>(V, 1) * 8));
        unsigned int *seg_start = c.vox_tmp.as<unsigned int>();
        hipLaunchKernelGGL(k_row_segments, g, dim3(256), 0, ctx->stream, kout.as<unsigned long long>(), flags, pos, m,
                           c.vox_key.as<unsigned long long>(), seg_start);
        if (V > 0) {
            // the rows of the last real voxel end where the padding segment starts (entry V of seg_start) or at m
            long long m_end = m;
            if (pad) {
                unsigned int s = 0;
                ME_TRY(copy_d2h(ctx, &s, seg_start + V, 4));
                ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                m_end = s;
            }
            TimerScope ts(ctx, "voxel");
            hipLaunchKernelGGL(k_vox_merge, dim3((unsigned int) ((V + 255) / 256)), dim3(256), 0, ctx->stream, rows_device,
                               perm.as<unsigned int>(), seg_start, V, m_end, c.vox_n.as<int>(), c.vox_mu.as<double>(),
                               c.vox_sigma.as<double>(), c.vox_entropy.as<double>());
        }
        ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ME_CHECK(ctx, hipGetLastError());
    }
    c.n_vox = V;
    c.vox_size = voxel_size;
    c.vox_raw = false;
    c.vox_valid = true;
    c.vox_merged = true;
    return ME_OK;
}

}  // namespace me
