// me_render.hip — the reference's colour renderers, fed from the arrays the metric kernels left on the device.
//
//   renderDistanceOnPointCloud            map_eval.cpp:586-607   (raw_rendered_dis_map.pcd, inlier_rendered_dis_map.pcd)
//   ColorPointCloudByMME(cloud, entropies) map_eval.cpp:686-735  (map_entropy.pcd, gt_entropy.pcd)
//   open3d::visualization::ColorMapJet     [Open3D ColorMap, upstream — restated, not in the reference tree]
//
// The reference repeats a serial KD-tree pass (computePointCloudDistance, :568-584) to colour the map by distance; the
// squared distances of the last me_nn1 are the same numbers (bit for bit), so nothing is searched again here.
#include <cmath>

#include "me_internal.hpp"

namespace me {

// ColorMap::Interpolate / ColorMapJet::JetBase / GetColor [Open3D, upstream]
__device__ __forceinline__ double jet_interpolate(double value, double y0, double x0, double y1, double x1) {
    if (value < x0) return y0;
    if (value > x1) return y1;
    return (value - x0) * (y1 - y0) / (x1 - x0) + y0;
}
__device__ __forceinline__ double jet_base(double value) {
    if (value <= -0.75) return 0.0;
    if (value <= -0.25) return jet_interpolate(value, 0.0, -0.75, 1.0, -0.25);
    if (value <= 0.25) return 1.0;
    if (value <= 0.75) return jet_interpolate(value, 1.0, 0.25, 0.0, 0.75);
    return 0.0;
}
__device__ __forceinline__ void jet_color(double value, double *rgb) {
    rgb[0] = jet_base(value * 2.0 - 1.5);
    rgb[1] = jet_base(value * 2.0 - 1.0);
    rgb[2] = jet_base(value * 2.0 - 0.5);
}

// one thread per (curve-ordered) query: colour + inlier flag scattered to the caller's cloud order
__global__ void k_render_distance(const SPoint *__restrict__ qsp, const double *__restrict__ d2s, long long n, double dis,
                                  double gate, int gate_strict, double *__restrict__ rgb, unsigned char *__restrict__ inlier) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long o = qsp[i].idx;
    const double d2 = d2s[i];
    double e = d2;
    if (e > dis) e = dis;            // (:591-595) squared distance against the unsquared threshold (sic)
    jet_color(e / dis, rgb + 3 * o);  // (:601-603)
    if (inlier) inlier[o] = (gate < 0) ? 1 : (gate_strict ? (d2 < gate) : (d2 <= gate));
}

// min / max over the non-zero entropies (:696-699); order-independent, so plain block partials + a final pass
__global__ void __launch_bounds__(256)
k_minmax_nonzero(const double *__restrict__ v, long long n, double *__restrict__ part) {
    double mn = INFINITY, mx = -INFINITY;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        const double x = v[i];
        if (x != 0.0) {
            mn = fmin(mn, x);
            mx = fmax(mx, x);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, o, 64));
        mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    __shared__ double s[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        s[2 * w] = mn;
        s[2 * w + 1] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmin(fmin(s[0], s[2]), fmin(s[4], s[6]));
        part[2 * blockIdx.x + 1] = fmax(fmax(s[1], s[3]), fmax(s[5], s[7]));
    }
}

// sorted order -> cloud order: entropy and validity flag (as a 32-bit flag for the scan)
__global__ void k_entropy_unpermute(const SPoint *__restrict__ sp, const double *__restrict__ ent_s,
                                    const unsigned char *__restrict__ val_s, long long n, double *__restrict__ ent_o,
                                    unsigned int *__restrict__ flag_o) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long o = sp[i].idx;
    ent_o[o] = ent_s[i];
    flag_o[o] = val_s[i] ? 1u : 0u;
}

// valid points only, in cloud order (:703-731)
__global__ void k_render_entropy(const double *__restrict__ xyz, const double *__restrict__ ent_o,
                                 const unsigned int *__restrict__ flag_o, const unsigned int *__restrict__ pos, long long n,
                                 double min_abs, double max_abs, double *__restrict__ xyz_out, double *__restrict__ rgb_out) {
    const long long o = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n || !flag_o[o]) return;
    const long long m = pos[o];
    const double epsilon = 1e-1;
    double ne = (fabs(ent_o[o]) - min_abs) / (max_abs - min_abs);               // (:714-715)
    const double mapped = log(ne + epsilon);                                    // (:719)
    ne = (mapped - log(epsilon)) / (log(1.0 + epsilon) - log(epsilon));         // (:721)
    jet_color(ne, rgb_out + 3 * m);
    xyz_out[3 * m] = xyz[3 * o];
    xyz_out[3 * m + 1] = xyz[3 * o + 1];
    xyz_out[3 * m + 2] = xyz[3 * o + 2];
}

static inline unsigned int blocks_for(long long n) { return (unsigned int) ((n + 255) / 256); }

int render_distance(me_ctx *ctx, int qslot, double dis, double gate, int gate_mode, double *rgb, uint8_t *inlier) {
    if (qslot < 0 || qslot > 1 || !rgb || !(dis > 0)) return ctx->fail(ME_ERR_ARG, "me_render_distance: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (q.nn_ref_slot < 0) return ctx->fail(ME_ERR_STATE, "no NN result for this slot (call me_nn1 first)");
    if (q.slab.axis >= 0 || ctx->shard_world > 1)
        return ctx->fail(ME_ERR_STATE, "me_render_distance: per-point outputs are not available in slab / shard mode");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = q.n;
    if (n == 0) return ME_OK;
    DevBuf &col = ctx->tmp[0], &inl = ctx->tmp[1];
    ME_CHECK(ctx, col.ensure((size_t) n * 24));
    ME_CHECK(ctx, inl.ensure((size_t) n));
    const double g = (gate < 0) ? -1.0 : (gate_mode == ME_GATE_LT_SQUARED ? gate * gate : gate);
    hipLaunchKernelGGL(k_render_distance, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(), q.nn_d2.as<double>(), n,
                       dis, g, gate_mode == ME_GATE_LT_SQUARED ? 1 : 0, col.as<double>(), inlier ? inl.as<unsigned char>() : nullptr);
    ME_TRY(copy_d2h(ctx, rgb, col.p, (size_t) n * 24));
    if (inlier) ME_TRY(copy_d2h(ctx, inlier, inl.p, (size_t) n));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

int render_entropy(me_ctx *ctx, int slot, double *xyz_out, double *rgb_out, long long capacity, long long *n_valid,
                   double *min_abs_out, double *max_abs_out) {
    if (slot < 0 || slot > 1 || !n_valid) return ctx->fail(ME_ERR_ARG, "me_render_entropy: bad argument");
    Cloud &c = ctx->cloud[slot];
    if (!c.mme_have) return ctx->fail(ME_ERR_STATE, "no MME result for this slot (call me_mme first; an upload or a transform discards it)");
    if (c.slab.axis >= 0 || ctx->shard_world > 1)
        return ctx->fail(ME_ERR_STATE, "me_render_entropy: per-point outputs are not available in slab / shard mode");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = c.n;
    *n_valid = 0;
    if (n == 0) return ME_OK;
    // --- range over the non-zero entropies ---
    const int nb = (int) std::min<long long>(1024, (n + 255) / 256);
    ME_CHECK(ctx, ctx->red.ensure((size_t) nb * 16));
    hipLaunchKernelGGL(k_minmax_nonzero, dim3(nb), dim3(256), 0, ctx->stream, c.mme_ent.as<double>(), n, ctx->red.as<double>());
    std::vector<double> part((size_t) nb * 2);
    ME_TRY(copy_d2h(ctx, part.data(), ctx->red.p, part.size() * 8));
    // --- cloud order + compaction offsets ---
    DevBuf &eo = ctx->tmp[0], &fl = ctx->tmp[1], &ps = ctx->tmp[2];
    ME_CHECK(ctx, eo.ensure((size_t) n * 8));
    ME_CHECK(ctx, fl.ensure((size_t) n * 4));
    ME_CHECK(ctx, ps.ensure((size_t) n * 4));
    hipLaunchKernelGGL(k_entropy_unpermute, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), c.mme_ent.as<double>(),
                       c.mme_val.as<unsigned char>(), n, eo.as<double>(), fl.as<unsigned int>());
    ME_TRY(exclusive_scan_u32(ctx, fl.as<unsigned int>(), ps.as<unsigned int>(), n));
    unsigned int last_pos = 0, last_flag = 0;
    ME_TRY(copy_d2h(ctx, &last_pos, ps.as<unsigned int>() + (n - 1), 4));
    ME_TRY(copy_d2h(ctx, &last_flag, fl.as<unsigned int>() + (n - 1), 4));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double mn = INFINITY, mx = -INFINITY;
    for (int b = 0; b < nb; ++b) {
        mn = std::fmin(mn, part[2 * b]);
        mx = std::fmax(mx, part[2 * b + 1]);
    }
    const double max_abs = std::fabs(mn), min_abs = std::fabs(mx);  // (:698-699: max_abs = |min element|, min_abs = |max element|)
    if (min_abs_out) *min_abs_out = min_abs;
    if (max_abs_out) *max_abs_out = max_abs;
    const long long m = (long long) last_pos + last_flag;
    *n_valid = m;
    if (!xyz_out && !rgb_out) return ME_OK;
    if (!xyz_out || !rgb_out) return ctx->fail(ME_ERR_ARG, "me_render_entropy: pass both xyz and rgb, or neither");
    if (capacity < m) return ctx->fail(ME_ERR_CAPACITY, "me_render_entropy: capacity too small");
    if (m == 0) return ME_OK;
    DevBuf &xo = ctx->tmp[3], &co = ctx->tmp[4];
    ME_CHECK(ctx, xo.ensure((size_t) m * 24));
    ME_CHECK(ctx, co.ensure((size_t) m * 24));
    hipLaunchKernelGGL(k_render_entropy, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, c.xyz.as<double>(), eo.as<double>(),
                       fl.as<unsigned int>(), ps.as<unsigned int>(), n, min_abs, max_abs, xo.as<double>(), co.as<double>());
    ME_TRY(copy_d2h(ctx, xyz_out, xo.p, (size_t) m * 24));
    ME_TRY(copy_d2h(ctx, rgb_out, co.p, (size_t) m * 24));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ME_CHECK(ctx, hipGetLastError());
    return ME_OK;
}

// ---- results computed elsewhere (the ranks of a distributed run) -> this context's sorted order ----
__global__ void k_permute_in(const SPoint *__restrict__ sp, long long n, const double *__restrict__ ent_o,
                             const unsigned char *__restrict__ val_o, double *__restrict__ ent_s, unsigned char *__restrict__ val_s,
                             int *__restrict__ idx_s) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long o = sp[i].idx;
    ent_s[i] = ent_o[o];
    if (val_s) val_s[i] = val_o[o];
    if (idx_s) idx_s[i] = -1;
}

int set_mme_result(me_ctx *ctx, int slot, const double *entropies, const uint8_t *valid) {
    if (slot < 0 || slot > 1 || !entropies || !valid) return ctx->fail(ME_ERR_ARG, "me_set_mme_result: bad argument");
    Cloud &c = ctx->cloud[slot];
    if (!c.uploaded || !c.index_valid) return ctx->fail(ME_ERR_STATE, "me_set_mme_result: cloud not uploaded");
    if (c.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_set_mme_result: the context must hold the whole cloud (no slab)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = c.n;
    DevBuf &eo = ctx->tmp[2], &vo = ctx->tmp[3];
    ME_CHECK(ctx, eo.ensure((size_t) n * 8));
    ME_CHECK(ctx, vo.ensure((size_t) n));
    ME_CHECK(ctx, c.mme_ent.ensure((size_t) n * 8));
    ME_CHECK(ctx, c.mme_val.ensure((size_t) n));
    ME_TRY(copy_h2d(ctx, eo.p, entropies, (size_t) n * 8));
    ME_TRY(copy_h2d(ctx, vo.p, valid, (size_t) n));
    hipLaunchKernelGGL(k_permute_in, dim3((unsigned int) ((n + 255) / 256)), dim3(256), 0, ctx->stream, c.sp.as<SPoint>(), n,
                       eo.as<double>(), vo.as<unsigned char>(), c.mme_ent.as<double>(), c.mme_val.as<unsigned char>(), (int *) nullptr);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    c.mme_have = true;
    return ME_OK;
}

int set_nn_result(me_ctx *ctx, int qslot, int rslot, const double *d2) {
    if (qslot < 0 || qslot > 1 || rslot < 0 || rslot > 1 || !d2) return ctx->fail(ME_ERR_ARG, "me_set_nn_result: bad argument");
    Cloud &q = ctx->cloud[qslot];
    if (!q.uploaded || !q.index_valid) return ctx->fail(ME_ERR_STATE, "me_set_nn_result: cloud not uploaded");
    if (q.slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_set_nn_result: the context must hold the whole cloud (no slab)");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = q.n;
    DevBuf &eo = ctx->tmp[2];
    ME_CHECK(ctx, eo.ensure((size_t) n * 8));
    ME_CHECK(ctx, q.nn_d2.ensure((size_t) n * 8));
    ME_CHECK(ctx, q.nn_idx.ensure((size_t) n * 4));
    ME_TRY(copy_h2d(ctx, eo.p, d2, (size_t) n * 8));
    hipLaunchKernelGGL(k_permute_in, dim3((unsigned int) ((n + 255) / 256)), dim3(256), 0, ctx->stream, q.sp.as<SPoint>(), n,
                       eo.as<double>(), (const unsigned char *) nullptr, q.nn_d2.as<double>(), (unsigned char *) nullptr, q.nn_idx.as<int>());
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    q.nn_ref_slot = rslot;  // (the neighbour indices are not known here: -1; the renderers and the statistics use d2 only)
    q.n_unres = 0;
    return ME_OK;
}

}  // namespace me
