// me_api.hip — the extern "C" surface declared in include/mapeval_hip.h (context, timers, call sequencing).
#include <cmath>
#include <cstring>

#include "me_internal.hpp"

static std::string g_create_error;

hipEvent_t me_ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}

void me_ctx::timer_begin(const char *name) {
    if (!timers_on) return;
    Pending p;
    p.name = name;
    p.a = get_event();
    p.b = get_event();
    (void) hipEventRecord(p.a, stream);
    pending.push_back(p);
}

void me_ctx::timer_end() {
    if (!timers_on || pending.empty()) return;
    // the most recent un-closed scope (scopes do not nest)
    (void) hipEventRecord(pending.back().b, stream);
}

void me_ctx::timers_collect() {
    if (pending.empty()) return;
    (void) hipStreamSynchronize(stream);
    for (auto &p : pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &t = timers[p.name];
            t.total_ms += ms;
            t.launches += 1;
        }
        event_pool.push_back(p.a);
        event_pool.push_back(p.b);
    }
    pending.clear();
}

namespace me {
__global__ void __launch_bounds__(64) k_mail(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, unsigned int bytes) {
    if (((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst) | bytes) & 7u) == 0) {
        for (unsigned int i = threadIdx.x; i < bytes / 8; i += 64)
            reinterpret_cast<unsigned long long *>(dst)[i] = reinterpret_cast<const unsigned long long *>(src)[i];
    } else {
        for (unsigned int i = threadIdx.x; i < bytes; i += 64) dst[i] = src[i];
    }
}

// Copies between caller (host) memory and the device, in one place.  copy_h2d is asynchronous on the context's stream for pinned
// sources and returns when a pageable source may be reused (the runtime stages it); copy_d2h returns when dst holds the data.
int copy_h2d(me_ctx *ctx, void *dst_device, const void *src_host, size_t bytes) {
    if (bytes == 0) return ME_OK;
    ME_CHECK(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ME_OK;
}

int copy_d2h(me_ctx *ctx, void *dst_host, const void *src_device, size_t bytes) {
    if (bytes == 0) return ME_OK;
    ME_CHECK(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int mail_post(me_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes == 0) return ME_OK;
    if (!ctx->mail_h) {
        void *h = nullptr, *d = nullptr;
        ME_CHECK(ctx, hipHostMalloc(&h, kMailBytes, hipHostMallocMapped));
        ME_CHECK(ctx, hipHostGetDevicePointer(&d, h, 0));
        ctx->mail_h = static_cast<unsigned char *>(h);
        ctx->mail_d = static_cast<unsigned char *>(d);
    }
    const size_t off = (ctx->mail_used + 15) & ~(size_t) 15;
    if (off + bytes > kMailBytes) return copy_d2h(ctx, host_dst, dev_src, bytes);  // (too big for the mailbox: the staged copy, at once)
    hipLaunchKernelGGL(k_mail, dim3(1), dim3(64), 0, ctx->stream, static_cast<const unsigned char *>(dev_src), ctx->mail_d + off,
                       (unsigned int) bytes);
    ctx->mail_pending.push_back({host_dst, off, bytes});
    ctx->mail_used = off + bytes;
    return ME_OK;
}

int mail_sync(me_ctx *ctx) {
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    // (a failed synchronisation delivers nothing, but the queue is emptied all the same: its entries point at the caller's locals)
    if (e == hipSuccess)
        for (const auto &m : ctx->mail_pending) std::memcpy(m.host, ctx->mail_h + m.off, m.bytes);
    ctx->mail_pending.clear();
    ctx->mail_used = 0;
    ME_CHECK(ctx, e);
    return ME_OK;
}

void mail_drop(me_ctx *ctx) {
    ctx->mail_pending.clear();
    ctx->mail_used = 0;
}
}  // namespace me

extern "C" {

int me_version(void) { return 100; }

static int stream_priority_for(bool main_lane) {
    int least = 0, greatest = 0;
    (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
    return main_lane ? greatest : least;  // (numerically lower = higher priority)
}

me_ctx *me_create(int device, int flags) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("me_create: no HIP device available (") + hipGetErrorString(e) +
                         "); libmapeval_hip has no CPU fallback";
        return nullptr;
    }
    if (device < 0 || device >= count) {
        g_create_error = "me_create: device ordinal out of range";
        return nullptr;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("me_create: hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    me_ctx *ctx = new me_ctx();
    ctx->device = device;
    ctx->borrow_device_input = (flags & ME_FLAG_BORROW_DEVICE_INPUT) != 0;
    ctx->morton_order = (flags & ME_FLAG_MORTON_ORDER) != 0;
    // The primary context's stream gets the highest dispatch priority, a twin's the lowest: when both lanes have kernels
    // queued, the main lane's (MME, 1-NN: the step's critical path) are dispatched first and the second lane's index / voxel
    // kernels fill in — 54.9 -> 53.4 ms per bench step (round 3, three runs each; the other two assignments measured slower).
    e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, stream_priority_for(true));
    if (e != hipSuccess) {
        g_create_error = std::string("me_create: hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

me_ctx *me_twin(me_ctx *ctx) {
    if (!ctx) return nullptr;
    if (ctx->is_twin) return ctx;
    if (ctx->twin) return ctx->twin;
    if (hipSetDevice(ctx->device) != hipSuccess) {
        ctx->fail(ME_ERR_HIP, "me_twin: hipSetDevice failed");
        return nullptr;
    }
    me_ctx *t = new me_ctx();
    t->device = ctx->device;
    t->is_twin = true;
    t->cloud.p[0] = ctx->cloud.p[0];
    t->cloud.p[1] = ctx->cloud.p[1];
    t->shard_rank = ctx->shard_rank;
    t->shard_world = ctx->shard_world;
    t->borrow_device_input = ctx->borrow_device_input;
    t->morton_order = ctx->morton_order;
    t->slab = ctx->slab;
    t->vox_hint = ctx->vox_hint;
    if (hipStreamCreateWithPriority(&t->stream, hipStreamNonBlocking, stream_priority_for(false)) != hipSuccess) {
        delete t;
        ctx->fail(ME_ERR_HIP, "me_twin: hipStreamCreate failed");
        return nullptr;
    }
    ctx->twin = t;
    return t;
}

void me_destroy(me_ctx *ctx) {
    if (!ctx) return;
    if (ctx->is_twin) return;  // twins belong to their primary context
    if (ctx->suite_worker && ctx->suite_worker_free) {  // (before the twin goes: the worker drives it)
        ctx->suite_worker_free(ctx->suite_worker);
        ctx->suite_worker = nullptr;
    }
    if (ctx->twin) {
        me_ctx *t = ctx->twin;
        ctx->twin = nullptr;
        t->is_twin = false;
        me_destroy(t);
    }
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending) {
        (void) hipEventDestroy(p.a);
        (void) hipEventDestroy(p.b);
    }
    for (auto e : ctx->event_pool) (void) hipEventDestroy(e);
    if (ctx->suite_event) (void) hipEventDestroy(static_cast<hipEvent_t>(ctx->suite_event));
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    if (ctx->mail_h) (void) hipHostFree(ctx->mail_h);
    delete ctx;
}

const char *me_last_error(me_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int me_set_shard(me_ctx *ctx, int rank, int world) {
    if (!ctx) return ME_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) return ctx->fail(ME_ERR_ARG, "me_set_shard: need 0 <= rank < world");
    ctx->shard_rank = rank;
    ctx->shard_world = world;
    if (ctx->twin) {
        ctx->twin->shard_rank = rank;
        ctx->twin->shard_world = world;
    }
    return ME_OK;
}

int me_upload_cloud(me_ctx *ctx, int slot, const double *xyz_host, int64_t n, const double *T, double cell_size) {
    if (!ctx) return ME_ERR_ARG;
    return me::cloud_upload(ctx, slot, xyz_host, false, n, T, cell_size);
}

int me_upload_cloud_device(me_ctx *ctx, int slot, const double *xyz_device, int64_t n, const double *T, double cell_size) {
    if (!ctx) return ME_ERR_ARG;
    return me::cloud_upload(ctx, slot, xyz_device, true, n, T, cell_size);
}

int me_upload_slab_device(me_ctx *ctx, int slot, const double *xyz_device, int64_t n, double cell_size) {
    if (!ctx) return ME_ERR_ARG;
    if (ctx->slab.axis < 0) return ctx->fail(ME_ERR_STATE, "me_upload_slab_device: call me_set_slab first");
    return me::cloud_upload(ctx, slot, xyz_device, true, n, nullptr, cell_size, true);
}

int64_t me_cloud_size(me_ctx *ctx, int slot) {
    if (!ctx || slot < 0 || slot > 1 || !ctx->cloud[slot].uploaded) return -1;
    return ctx->cloud[slot].n;
}

int me_download_cloud(me_ctx *ctx, int slot, double *xyz_host) {
    if (!ctx) return ME_ERR_ARG;
    if (slot < 0 || slot > 1 || !xyz_host) return ctx->fail(ME_ERR_ARG, "me_download_cloud: bad argument");
    me::Cloud &c = ctx->cloud[slot];
    if (!c.uploaded) return ctx->fail(ME_ERR_STATE, "cloud not uploaded");
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    ME_TRY(me::copy_d2h(ctx, xyz_host, c.xyz.p, (size_t) c.n * 24));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int me_voxel_downsample(me_ctx *ctx, int slot, double voxel_size, int64_t *n_out) {
    if (!ctx) return ME_ERR_ARG;
    long long n = 0;
    const int rc = me::voxel_downsample(ctx, slot, voxel_size, &n);
    if (n_out) *n_out = n;
    return rc;
}

int me_transform_cloud(me_ctx *ctx, int slot, const double *T) {
    if (!ctx) return ME_ERR_ARG;
    return me::cloud_transform(ctx, slot, T);
}

int me_nn1(me_ctx *ctx, int query_slot, int ref_slot, int32_t *idx, double *d2) {
    if (!ctx) return ME_ERR_ARG;
    ME_TRY(me::nn_search(ctx, query_slot, ref_slot));
    if (idx || d2) return me::nn_fetch(ctx, query_slot, idx, d2);
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ME_OK;
}

int me_icp_p2p_sums(me_ctx *ctx, int query_slot, double max_distance, me_icp_sums *out) {
    if (!ctx) return ME_ERR_ARG;
    return me::icp_p2p_sums(ctx, query_slot, max_distance, out);
}

int me_set_normals(me_ctx *ctx, int slot, const double *normals) {
    if (!ctx) return ME_ERR_ARG;
    return me::set_normals(ctx, slot, normals);
}

int me_get_normals(me_ctx *ctx, int slot, double *normals) {
    if (!ctx) return ME_ERR_ARG;
    return me::get_normals(ctx, slot, normals);
}

int me_estimate_normals(me_ctx *ctx, int slot, int knn, double *normals, int32_t *knn_idx, double *knn_d2) {
    if (!ctx) return ME_ERR_ARG;
    return me::estimate_normals(ctx, slot, knn, normals, knn_idx, knn_d2);
}

int me_gicp_covariances(me_ctx *ctx, int slot, double epsilon, double *cov) {
    if (!ctx) return ME_ERR_ARG;
    return me::gicp_covariances(ctx, slot, epsilon, cov);
}

int me_get_covariances(me_ctx *ctx, int slot, double *cov) {
    if (!ctx) return ME_ERR_ARG;
    return me::get_covariances(ctx, slot, cov);
}

int me_icp_lsq_sums(me_ctx *ctx, int query_slot, int mode, double max_distance, me_icp_lsq *out) {
    if (!ctx) return ME_ERR_ARG;
    return me::icp_lsq_sums(ctx, query_slot, mode, max_distance, out);
}

int me_render_distance(me_ctx *ctx, int query_slot, double dis, double gate, int gate_mode, double *rgb, uint8_t *inlier) {
    if (!ctx) return ME_ERR_ARG;
    return me::render_distance(ctx, query_slot, dis, gate, gate_mode, rgb, inlier);
}

int me_render_entropy(me_ctx *ctx, int slot, double *xyz, double *rgb, int64_t capacity, int64_t *n_valid, double *min_abs,
                      double *max_abs) {
    if (!ctx) return ME_ERR_ARG;
    long long nv = 0;
    const int rc = me::render_entropy(ctx, slot, xyz, rgb, capacity, &nv, min_abs, max_abs);
    if (n_valid) *n_valid = nv;
    return rc;
}

int me_nn_partial_sums(me_ctx *ctx, int query_slot, double gate, int gate_mode, const double trunc[5], me_nn_partial *out) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_partial(ctx, query_slot, gate, gate_mode, trunc, out);
}

int me_nn_sigma_sums(me_ctx *ctx, int query_slot, double gate, int gate_mode, const double mean[5], double sigma_num[5]) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_sigma(ctx, query_slot, gate, gate_mode, mean, sigma_num);
}

void me_nn_finalize(const me_nn_partial *t, const double sigma_num[5], int64_t n_src_total, me_nn_stats_out *out) {
    // map_eval.cpp:1125-1144.  C == 0 gives 0/0 = NaN for mean / rmse / sigma, exactly as the reference.
    const double C = (double) t->n_corr;
    out->n_src = n_src_total;
    out->n_corr = t->n_corr;
    for (int k = 0; k < 5; ++k) {
        out->mean[k] = t->sum_d[k] / C;                                // mean_vec /= points_set.size()   (:1125)
        out->rmse[k] = std::sqrt(t->sum_d2[k] / C);                    // sqrt(rmse_vec / size)           (:1126,:1131)
        out->fitness[k] = (double) t->n_inl[k] * 1.0 / (double) n_src_total;  // number / source.size()   (:1130)
        out->sigma[k] = std::sqrt(sigma_num[k] / C);                   // (:1137-1138)
        out->number[k] = (double) t->n_inl[k];
    }
    out->mean_nn_dist = t->sum_sqrt_all / (double) n_src_total;        // sum / N (:1429)
}

int me_nn_stats(me_ctx *ctx, int query_slot, double gate, int gate_mode, const double trunc[5], me_nn_stats_out *out) {
    if (!ctx) return ME_ERR_ARG;
    if (!out) return ctx->fail(ME_ERR_ARG, "me_nn_stats: out is NULL");
    if (ctx->shard_world != 1 || ctx->slab.axis >= 0)
        return ctx->fail(ME_ERR_STATE, "me_nn_stats is the single-GPU one-shot; use me_nn_partial_sums / me_nn_sigma_sums when sharded");
    me_nn_partial p;
    ME_TRY(me::nn_partial(ctx, query_slot, gate, gate_mode, trunc, &p));
    double mean[5], sig[5];
    for (int k = 0; k < 5; ++k) mean[k] = p.sum_d[k] / (double) p.n_corr;
    ME_TRY(me::nn_sigma(ctx, query_slot, gate, gate_mode, mean, sig));
    me_nn_finalize(&p, sig, ctx->cloud[query_slot].n, out);
    return ME_OK;
}

int me_chamfer(me_ctx *ctx, double *cd) {
    if (!ctx) return ME_ERR_ARG;
    if (!cd) return ctx->fail(ME_ERR_ARG, "me_chamfer: cd is NULL");
    if (ctx->shard_world != 1 || ctx->slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_chamfer is single-GPU; use the partial-sum calls when sharded");
    const double tr[5] = {0, 0, 0, 0, 0};
    me_nn_partial a, b;
    ME_TRY(me::nn_search(ctx, ME_SLOT_EST, ME_SLOT_GT));
    ME_TRY(me::nn_partial(ctx, ME_SLOT_EST, -1.0, 0, tr, &a));
    ME_TRY(me::nn_search(ctx, ME_SLOT_GT, ME_SLOT_EST));
    ME_TRY(me::nn_partial(ctx, ME_SLOT_GT, -1.0, 0, tr, &b));
    *cd = a.sum_sqrt_all / (double) ctx->cloud[ME_SLOT_EST].n + b.sum_sqrt_all / (double) ctx->cloud[ME_SLOT_GT].n;  // (:1429)
    return ME_OK;
}

int me_mme(me_ctx *ctx, int slot, double radius, int min_k, double *entropies, uint8_t *valid, double *sum_H, int64_t *n_valid) {
    if (!ctx) return ME_ERR_ARG;
    long long nv = 0;
    const int rc = me::mme_run(ctx, slot, radius, min_k, entropies, valid, sum_H, &nv);
    if (n_valid) *n_valid = nv;
    return rc;
}

int me_mme_fetch(me_ctx *ctx, int slot, double *entropies, uint8_t *valid) {
    if (!ctx) return ME_ERR_ARG;
    return me::mme_fetch(ctx, slot, entropies, valid);
}

int me_voxel_gaussians(me_ctx *ctx, int slot, double voxel_size, int32_t *keys, int32_t *npts, double *mu, double *sigma,
                       double *entropy, int64_t *n_voxels) {
    if (!ctx) return ME_ERR_ARG;
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    if (ctx->cloud[slot].slab.axis >= 0)
        return ctx->fail(ME_ERR_STATE, "me_voxel_gaussians: in slab mode use me_voxel_partials and merge across ranks");
    ME_TRY(me::voxel_build(ctx, slot, voxel_size, false));
    return me::voxel_export(ctx, slot, keys, npts, mu, sigma, entropy, n_voxels);
}

int me_voxel_partials(me_ctx *ctx, int slot, double voxel_size, int32_t *keys, int32_t *npts, double *mu, double *m2,
                      int64_t *n_voxels) {
    if (!ctx) return ME_ERR_ARG;
    if (slot < 0 || slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    ME_TRY(me::voxel_build(ctx, slot, voxel_size, true));
    return me::voxel_export(ctx, slot, keys, npts, mu, m2, nullptr, n_voxels);
}

int me_set_slab(me_ctx *ctx, int axis, double lo, double hi, double halo) {
    if (!ctx) return ME_ERR_ARG;
    if (axis < 0) {
        ctx->slab = me::SlabView{-1, 0, 0, 0, 0};
        if (ctx->twin) ctx->twin->slab = ctx->slab;
        return ME_OK;
    }
    if (axis > 2 || !(lo < hi) || !(halo >= 0)) return ctx->fail(ME_ERR_ARG, "me_set_slab: need axis in 0..2, lo < hi, halo >= 0");
    ctx->slab = me::SlabView{axis, lo, hi, lo - halo, hi + halo};
    if (ctx->twin) ctx->twin->slab = ctx->slab;
    return ME_OK;
}

int me_nn_unresolved(me_ctx *ctx, int query_slot, double *xyz_device, double *d2_device, int64_t capacity, int64_t *count) {
    if (!ctx) return ME_ERR_ARG;
    long long c = 0;
    const int rc = me::nn_unresolved(ctx, query_slot, xyz_device, d2_device, capacity, &c);
    if (count) *count = c;
    return rc;
}

int me_nn_points(me_ctx *ctx, int ref_slot, const double *xyz_device, int64_t m, double *d2_device) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_points(ctx, ref_slot, xyz_device, m, d2_device, false);
}

int me_nn_points_bounded(me_ctx *ctx, int ref_slot, const double *xyz_device, int64_t m, double *d2_inout_device) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_points(ctx, ref_slot, xyz_device, m, d2_inout_device, true);
}

int me_nn_points_covered(me_ctx *ctx, int ref_slot, const double *xyz_device, int64_t m, double *d2_inout_device, int axis,
                         const double *covered_device) {
    if (!ctx) return ME_ERR_ARG;
    if (axis < 0 || axis > 2) return ctx->fail(ME_ERR_ARG, "me_nn_points_covered: axis must be 0, 1 or 2");
    if (m > 0 && !covered_device) return ctx->fail(ME_ERR_ARG, "me_nn_points_covered: covered intervals missing");
    return me::nn_points(ctx, ref_slot, xyz_device, m, d2_inout_device, true, axis, covered_device);
}

int me_nn_fetch(me_ctx *ctx, int query_slot, int32_t *idx, double *d2) {
    if (!ctx) return ME_ERR_ARG;
    if (query_slot < 0 || query_slot > 1) return ctx->fail(ME_ERR_ARG, "bad slot");
    return me::nn_fetch(ctx, query_slot, idx, d2);
}

int me_slab_points(me_ctx *ctx, int slot, int64_t *orig_index, uint8_t *owned, int64_t capacity, int64_t *count) {
    if (!ctx) return ME_ERR_ARG;
    long long c = 0;
    const int rc = me::slab_points(ctx, slot, orig_index, owned, capacity, &c);
    if (count) *count = c;
    return rc;
}

int me_set_mme_result(me_ctx *ctx, int slot, const double *entropies, const uint8_t *valid) {
    if (!ctx) return ME_ERR_ARG;
    return me::set_mme_result(ctx, slot, entropies, valid);
}

int me_set_nn_result(me_ctx *ctx, int query_slot, int ref_slot, const double *d2) {
    if (!ctx) return ME_ERR_ARG;
    return me::set_nn_result(ctx, query_slot, ref_slot, d2);
}

int me_nn_patch(me_ctx *ctx, int query_slot, const double *d2_device, int64_t count) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_patch(ctx, query_slot, d2_device, count);
}

int me_nn_cross_message(me_ctx *ctx, double *msg_device, int64_t capacity, int64_t n_local_est, int64_t n_local_gt, int64_t counts[2]) {
    if (!ctx) return ME_ERR_ARG;
    long long c[2] = {0, 0};
    const int rc = me::nn_cross_message(ctx, msg_device, capacity, n_local_est, n_local_gt, c);
    if (counts) {
        counts[0] = c[0];
        counts[1] = c[1];
    }
    return rc;
}

int me_nn_cross_answer(me_ctx *ctx, const double *gathered_device, int world, int64_t capacity, int own_rank, int dir_mask, int axis,
                       const double *cuts, double halo, double *d2_device) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_cross_answer(ctx, gathered_device, world, capacity, own_rank, dir_mask, axis, cuts, halo, d2_device);
}

int me_nn_cross_patch(me_ctx *ctx, const double *d2_reduced_device, int64_t capacity, int own_rank) {
    if (!ctx) return ME_ERR_ARG;
    return me::nn_cross_patch(ctx, d2_reduced_device, capacity, own_rank);
}

int me_transform_points_device(me_ctx *ctx, double *xyz_device, int64_t n, const double *T) {
    if (!ctx) return ME_ERR_ARG;
    return me::transform_points_device(ctx, xyz_device, n, T);
}

int me_halo_pack_device(me_ctx *ctx, const double *xyz_device, int64_t n, int axis, const double *cuts, int world, double halo,
                        double *out_device, int64_t capacity, int64_t *counts) {
    if (!ctx) return ME_ERR_ARG;
    long long c[64];
    if (world < 1 || world > 64 || !counts) return ctx->fail(ME_ERR_ARG, "me_halo_pack_device: need 1 <= world <= 64 and counts");
    const int rc = me::halo_pack(ctx, xyz_device, n, axis, cuts, world, halo, out_device, capacity, c);
    for (int k = 0; k < world; ++k) counts[k] = c[k];
    return rc;
}

int me_halo_pack_tagged_device(me_ctx *ctx, const double *xyz_device, int64_t n, int axis, const double *cuts, int world, double halo,
                               double *out_device, int64_t *tags_device, int64_t tag_base, int64_t capacity, int64_t *counts) {
    if (!ctx) return ME_ERR_ARG;
    long long c[64];
    if (world < 1 || world > 64 || !counts) return ctx->fail(ME_ERR_ARG, "me_halo_pack_tagged_device: need 1 <= world <= 64 and counts");
    const int rc = me::halo_pack(ctx, xyz_device, n, axis, cuts, world, halo, out_device, capacity, c,
                                 reinterpret_cast<long long *>(tags_device), (long long) tag_base);
    for (int k = 0; k < world; ++k) counts[k] = c[k];
    return rc;
}

int me_lattice_histograms_device(me_ctx *ctx, const double *xyz_device, int64_t n, int e0, int32_t *level, int64_t origin_bin[3],
                                 int64_t neg_inf[3], uint32_t *hist_device) {
    if (!ctx) return ME_ERR_ARG;
    int lv = 0;
    long long o[3] = {0, 0, 0}, ni[3] = {0, 0, 0};
    if (!level || !origin_bin || !neg_inf) return ctx->fail(ME_ERR_ARG, "me_lattice_histograms_device: NULL output");
    const int rc = me::lattice_histograms(ctx, xyz_device, n, e0, &lv, o, ni, hist_device);
    *level = lv;
    for (int a = 0; a < 3; ++a) {
        origin_bin[a] = o[a];
        neg_inf[a] = ni[a];
    }
    return rc;
}

int me_lattice_messages_device(me_ctx *ctx, const double *xyz_a_device, int64_t n_a, const double *xyz_b_device, int64_t n_b, int clouds, int e0,
                               int64_t *msg_device) {
    if (!ctx) return ME_ERR_ARG;
    const double *x[2] = {xyz_a_device, xyz_b_device};
    const long long n[2] = {(long long) n_a, (long long) n_b};
    return me::lattice_messages(ctx, x, n, clouds, e0, reinterpret_cast<long long *>(msg_device));
}

int me_lattice_plan_device(me_ctx *ctx, const int64_t *msgs_device, int world, int clouds, double halo, int e0, int64_t *out) {
    if (!ctx) return ME_ERR_ARG;
    static_assert(sizeof(long long) == sizeof(int64_t), "int64_t is long long here");
    return me::lattice_plan(ctx, reinterpret_cast<const long long *>(msgs_device), world, clouds, halo, e0, reinterpret_cast<long long *>(out));
}

int me_voxel_partial_rows_device(me_ctx *ctx, int slot, double voxel_size, double *rows_device, int64_t capacity, int64_t *n_rows) {
    if (!ctx) return ME_ERR_ARG;
    long long n = 0;
    const int rc = me::voxel_rows_device(ctx, slot, voxel_size, rows_device, capacity, &n);
    if (n_rows) *n_rows = n;
    return rc;
}

int me_voxel_merge_device(me_ctx *ctx, int slot, double voxel_size, const double *rows_device, int64_t n_rows) {
    if (!ctx) return ME_ERR_ARG;
    return me::voxel_merge(ctx, slot, voxel_size, rows_device, n_rows);
}

int me_awd_scs(me_ctx *ctx, double voxel_size, int min_pts, int scs_radius, double *rows, double *w_sorted, int64_t *n_rows,
               double *awd, double *scs, int64_t counts[3]) {
    if (!ctx) return ME_ERR_ARG;
    return me::awd_scs(ctx, voxel_size, min_pts, scs_radius, rows, w_sorted, n_rows, awd, scs, counts);
}

int me_w2_batch(me_ctx *ctx, const double *mu1, const double *sigma1, const int32_t *n1, const double *mu2,
                const double *sigma2, const int32_t *n2, int64_t count, double *w) {
    if (!ctx) return ME_ERR_ARG;
    return me::w2_batch(ctx, mu1, sigma1, n1, mu2, sigma2, n2, count, w);
}

int me_scs_table(me_ctx *ctx, const int32_t *keys, const double *w, int64_t n, int scs_radius, double *scs) {
    if (!ctx) return ME_ERR_ARG;
    return me::scs_table(ctx, keys, w, n, scs_radius, scs);
}

int me_set_voxel_hint(me_ctx *ctx, double voxel_size) {
    if (!ctx) return ME_ERR_ARG;
    if (!(voxel_size >= 0)) return ctx->fail(ME_ERR_ARG, "me_set_voxel_hint: voxel_size must be >= 0 (0: off)");
    me_ctx *primary = ctx;
    primary->vox_hint = voxel_size;
    if (primary->twin) primary->twin->vox_hint = voxel_size;
    return ME_OK;
}

int me_timers_enable(me_ctx *ctx, int on) {
    if (!ctx) return ME_ERR_ARG;
    ctx->timers_collect();
    ctx->timers_on = on != 0;
    return ME_OK;
}

int me_timers_reset(me_ctx *ctx) {
    if (!ctx) return ME_ERR_ARG;
    ctx->timers_collect();
    ctx->timers.clear();
    ctx->nn_fallback = ctx->nn_queries = 0;
    ctx->mme_pairs = 0;
    ctx->mme_refined = 0;
    if (ctx->nn1_dbg_buf.p) (void) hipMemsetAsync(ctx->nn1_dbg_buf.p, 0, 128, ctx->stream);
    return ME_OK;
}

int me_timer_get(me_ctx *ctx, const char *name, double *total_ms, int64_t *launches) {
    if (!ctx || !name) return ME_ERR_ARG;
    ctx->timers_collect();
    if (std::strcmp(name, "nn_fallback_queries") == 0 || std::strcmp(name, "nn_queries") == 0) {
        // counters, not timers: 1-NN queries that needed the octree pass / all 1-NN queries since the last reset
        const long long v = name[3] == 'f' ? ctx->nn_fallback : ctx->nn_queries;
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = v;
        return ME_OK;
    }
    if (std::strcmp(name, "mme_pairs") == 0) {  // accepted (query, neighbour) pairs of the MME launches since the last reset
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = ctx->mme_pairs;
        return ME_OK;
    }
    if (std::strcmp(name, "mme_refined") == 0) {  // queries recomputed by k_mme_refine (thin neighbourhoods) since the last reset; counted always
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = ctx->mme_refined;
        return ME_OK;
    }
    if (std::strncmp(name, "nn1_", 4) == 0 && std::strlen(name) > 4) {
        // counters of the octree walk since the last reset (collected while timers are on): nodes opened, leaf cells scanned,
        // points scanned, the longest chain (opened + scanned) of one query
        // ... and of k_nn_far: nodes opened, points scanned, the longest chain of one query
        static const char *names[12] = {"nn1_opened", "nn1_scans", "nn1_points", "nn1_max_opened", "nn1_far", "nn1_far_opened", "nn1_far_points", "nn1_far_max",
                                         "nn1_wave_max_10ns", "nn1_wave_sum_10ns", "nn1_max_run", "nn1_waves"};
        for (int k = 0; k < 12; ++k)
            if (std::strcmp(name, names[k]) == 0) {
                unsigned long long v = 0;
                if (ctx->nn1_dbg_buf.p) {
                    (void) hipDeviceSynchronize();
                    (void) me::copy_d2h(ctx, &v, ctx->nn1_dbg_buf.as<unsigned long long>() + k, 8);
                }
                if (total_ms) *total_ms = 0.0;
                if (launches) *launches = (int64_t) v;
                return ME_OK;
            }
    }
    auto it = ctx->timers.find(name);
    if (total_ms) *total_ms = it == ctx->timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == ctx->timers.end() ? 0 : it->second.launches;
    return ME_OK;
}

}  // extern "C"
