// me_internal.hpp — shared declarations of libmapeval_hip.so (gfx950 only; no CPU fallback anywhere).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/mapeval_hip.h"

namespace me {

constexpr int kFan = 8;        // children per octree node
constexpr int kMaxLevels = 17; // octree levels above the leaf cells (taken-masks: 2 x 64 bits)
constexpr int kMortonBits = 21;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool owned = true;  // false: `p` is the caller's memory (ME_FLAG_BORROW_DEVICE_INPUT), read-only for the library
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    hipError_t ensure(size_t b) {
        if (owned && b <= bytes && p) return hipSuccess;
        release();
        if (b == 0) b = 16;
        hipError_t e = hipMalloc(&p, b);
        if (e == hipSuccess) bytes = b;
        else p = nullptr;
#ifdef ME_POISON_ALLOC  // debug builds (make EXTRA=-DME_POISON_ALLOC): fresh device memory is filled with a pattern, so that a kernel
        // which reads a buffer before anything wrote it shows at once (hipMalloc hands out zeroed pages in a fresh process and
        // recycled, dirty ones after a hipFree: such a bug appears only when a second context is created in a long-lived process)
        if (e == hipSuccess) (void) hipMemset(p, 0xCD, b);
#endif
        return e;
    }
    void release() {
        if (p && owned) (void) hipFree(p);
        p = nullptr;
        bytes = 0;
        owned = true;
    }
    void borrow(void *ptr, size_t b) {
        release();
        p = ptr;
        bytes = b;
        owned = false;
    }
    // before the library writes into the buffer: a borrowed one becomes a private copy
    hipError_t make_owned(hipStream_t stream) {
        if (owned) return hipSuccess;
        void *src = p;
        const size_t b = bytes;
        p = nullptr;
        bytes = 0;
        owned = true;
        hipError_t e = ensure(b);
        if (e == hipSuccess) e = hipMemcpyAsync(p, src, b, hipMemcpyDeviceToDevice, stream);
        return e;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
    // for kernels that WRITE the buffer: a borrowed buffer (the caller's memory, ME_FLAG_BORROW_DEVICE_INPUT) yields nullptr — a writer
    // that forgot ensure() / make_owned() faults at once instead of changing the caller's cloud
    template <class T>
    T *as_mut() const { return owned ? reinterpret_cast<T *>(p) : nullptr; }
};

// Sorted point: xyz in fp64 + original index (bit pattern of an int64 in w).
struct alignas(32) SPoint {
    double x, y, z;
    long long idx;
};

// Sparse octree over the Morton prefixes of the sorted points (the general 1-NN path).  Level 0 = the occupied cells of
// the 1-NN grid (each a contiguous run of `sp`), level l+1 = their unique 3-bit-shorter prefixes, up to a single root.
// A node's children are contiguous on the level below; every node's box lies inside its octree cell, so boxes of one
// level never straddle a Morton seam (a fixed-count grouping of consecutive points does, and was measured to cost
// ~150 node visits per query instead of ~20).
struct alignas(32) ONode {
    float lo[3], hi[3];   // tight fp32 box of the points below, rounded OUTWARD from the fp64 extent
    unsigned int begin;   // first child on the level below (level 0: first point); the node after it holds the end
    unsigned int parent;  // index on the level above
};
struct OctView {
    const ONode *nodes;
    const unsigned int *pbegin;  // first sorted point below every node (indexed like `nodes`; a level's terminator holds n)
    int n_levels;  // level 0 = leaf cells ... level n_levels-1 = root (1 node)
    long long count[kMaxLevels];
    long long off[kMaxLevels];  // offset of each level inside `nodes` (each level stores count + 1 records)
};

struct GridView {
    const unsigned long long *cell_code;  // unique cell Morton codes, in the order of the sorted points [n_cells]
    const unsigned int *cell_start;       // [n_cells + 1] offsets into the sorted points
    const unsigned long long *hkeys;      // open-addressing table, EMPTY = ~0ull
    const unsigned int *hvals;            // cell index
    unsigned int hmask;
    long long n_cells;
    int shift;                            // fine Morton code >> (3*shift) = cell code
};

struct GridTable {
    DevBuf cell_code, cell_start, hkeys, hvals;
    int shift = -1;
};

// Spatial slab of a context (multi-GPU): a point is OWNED iff lo <= p[axis] < hi; [reg_lo, reg_hi) = slab + halo is
// everything the context holds.  axis < 0: no slab, every point is owned.  +-inf faces never fail a test.
struct SlabView {
    int axis;
    double lo, hi;
    double reg_lo, reg_hi;
};
#ifdef __HIPCC__
__device__ __forceinline__ bool slab_owned(const SlabView &s, double x, double y, double z) {
    if (s.axis < 0) return true;
    const double v = s.axis == 0 ? x : (s.axis == 1 ? y : z);
    return v >= s.lo && v < s.hi;
}
__device__ __forceinline__ double slab_face_distance(const SlabView &s, double x, double y, double z) {
    if (s.axis < 0) return INFINITY;
    const double v = s.axis == 0 ? x : (s.axis == 1 ? y : z);
    return fmin(v - s.reg_lo, s.reg_hi - v);  // distance to the nearest face beyond which this context holds nothing
}
#endif

// Morton frame of a cloud, as the kernels need it to place a foreign point into this cloud's grid
struct FrameView {
    double ox, oy, oz;  // origin (bbox min)
    double fine_h;      // edge of the finest (21-bit) cell
};

// sort key layout of the one-pass voxel build's run records (me_vox_rows.hpp)
struct VoxPack {
    double vs;
    int min_x, min_y, min_z;   // smallest voxel index of the cloud's bounding box per axis
    int bits_y, bits_z;        // bits of the y / z index ranges (x takes what is left)
    int pos_bits;              // bits below the compact voxel key: row (i >> 6) and the run inside the row (6 bits)
    unsigned long long sentinel;  // compact key of the halo points (slab mode): above every real one
};

struct Cloud {
    // run records of the one-pass voxel build, emitted by the index build's gather when the context carries a voxel-size hint
    // (me_run_suite_from): slot keys | populations | sums, the counter block [range error, appended records]
    DevBuf vox_rec_key, vox_rec_n, vox_rec_s, vox_rec_cnt;
    bool vox_rec_valid = false;
    long long vox_rec_cap = 0;
    VoxPack vox_rec_pack{};
    bool sort_pairs_hint = false;  // the keys-only sort had to cut the sort depth and the cloud was dense: use the pair sort (me_index.hip)
    long long hint_n = 0;          // ... for a cloud of this shape only (same count, lattice depth and cell edge)
    int hint_shift = 0;
    double hint_cell_h = 0;
    long long n = 0;        // points held (slab mode: owned + halo)
    long long n_total = 0;  // points the caller passed to the upload
    SlabView slab{-1, 0, 0, 0, 0};
    DevBuf nn_unres;        // slab mode: sorted positions of the not-yet-global 1-NN results
    DevBuf slab_orig;       // slab mode, filtered upload: int32[n] position of every kept point in the uploaded array
    bool slab_identity = true;  // ... or the identity (me_upload_slab_device, no filter pass)
    long long n_unres = 0;
    bool uploaded = false;
    DevBuf xyz;  // double[n][3] original order, after the optional transform
    // Morton frame
    double origin[3] = {0, 0, 0};
    double bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};
    double cell_size_req = 0;  // cell size the caller asked for (<= 0: automatic)
    double cell_h = 0;   // radius-grid cell edge
    double fine_h = 0;   // cell_h / 2^shift
    int shift = 0;
    bool index_valid = false;
    DevBuf codes;  // uint64[n] fine Morton code of every sorted point (the points are sorted along the Hilbert curve)
    DevBuf sp;     // SPoint[n] sorted
    // sparse octree (general 1-NN path)
    OctView oct{};
    bool oct_deferred = false;  // the index is valid but its octree is still to be built (cloud_finish_octree)
    int oct_nn_shift = 0;       // level of the octree's leaves (the 1-NN grid's)
    DevBuf oct_nodes;
    DevBuf oct_pbegin;
    // cell tables: `grid` at the radius level (MME), `nn_grid` at the level whose occupied cells hold ~16 points
    // (1-NN fast path); they share storage when the two levels coincide
    GridTable grid_tab, nn_tab;
    GridView grid{}, nn_grid{};
    // the levels strictly between the 1-NN grid and the radius grid (dense clouds only: at 10^4 pts/m^2 the 1-NN cells are 2.5 cm and
    // the radius cells 10 cm): the 1-NN cascade passes what a level leaves unresolved to the next coarser one
    static constexpr int kMaxMid = 4;
    GridTable mid_tab[kMaxMid];
    GridView mid_grid[kMaxMid]{};
    int n_mid = 0;
    long long level_unique[kMortonBits + 1] = {0};  // occupied cells per Morton level
    // last NN result with this cloud as the query (sorted query order)
    DevBuf nn_d2, nn_idx, nn_list;
    int nn_ref_slot = -1;
    // voxel table (ascending key order)
    double vox_size = 0;
    long long n_vox = 0;
    bool vox_valid = false, vox_raw = false;
    bool vox_merged = false;  // the table is the cross-rank merge of partials (me_voxel_merge_device): complete on every rank
    DevBuf mme_ent, mme_val;  // last me_mme of this cloud, sorted order: entropy (0 where invalid), validity byte
    bool mme_have = false;
#ifdef ME_AB  // measurement build only (profiles/ab/me_mme7.hip): digit features of the matrix-pipe MME kernel
    DevBuf mme_feat;        // [n_chunks][80 columns][16 candidates] signed bytes
    DevBuf mme_cell_chunk;  // uint32[n_cells + 1] first chunk of every cell
    bool mme_feat_valid = false, mme_feat_usable = false;
    double mme_feat_radius = 0;
    long long mme_fx_origin[3] = {0, 0, 0};
    int mme_fx_scale = 0;
#endif
    DevBuf vox_tmp;    // build scratch (segment starts when they outgrow the shared scratch)
    DevBuf vox_key;    // uint64[V] packed key
    DevBuf vox_n;      // int32[V]
    DevBuf vox_mu;     // double[V][3]
    DevBuf vox_sigma;  // double[V][9] as stored by the reference
    DevBuf vox_entropy;
    // per-point attributes of the registration path, ORIGINAL point order (me_reg.hip)
    DevBuf normals;  // double[n][3]
    DevBuf cov;      // double[n][9] generalized-ICP covariances
    bool have_normals = false, have_cov = false;
};

struct TimerRec {
    double total_ms = 0;
    long long launches = 0;
};

}  // namespace me

struct me_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // The clouds live in `own_cloud` of the PRIMARY context; a twin lane (me_twin) points at the same two objects but has
    // its own stream, scratch and timers, so that two host threads can drive independent work concurrently.
    me::Cloud own_cloud[2];
    struct CloudSet {
        me::Cloud *p[2];
        me::Cloud &operator[](int i) const { return *p[i]; }
    } cloud;
    me_ctx *twin = nullptr;  // owned by the primary context
    // called by cloud_build_index right after its radix sort has been QUEUED on the context's stream (me_run_suite_from's second lane
    // uses it to let the main lane order its MME behind the sort: me_suite.hip)
    void (*sort_hook)(void *arg, hipStream_t stream) = nullptr;
    void *sort_hook_arg = nullptr;
    // me_run_suite_from's second-lane host thread (me_suite.hip: LaneWorker), created on first use, joined by me_destroy
    void *suite_event = nullptr;  // hipEvent_t of me_run_suite_from's "sort queued" hand-over, created on first use
    void *suite_worker = nullptr;
    void (*suite_worker_free)(void *) = nullptr;
    // Small device -> host results (sums, counts, the level histogram) go through a pinned, device-mapped MAILBOX written by a
    // one-wavefront kernel (me::mail_post / me::mail_sync, me_api.hip; round 4).  hipMemcpyAsync to pageable host memory is a blit
    // kernel of ONE 1024-thread workgroup: it needs 16 free wave slots on one CU at once, and while the other lane's k_nn_grid /
    // k_mme3 keeps every CU full that took 3 - 5.6 ms per read on the critical path (profiles/r04_timeline_two_lane.txt).
    unsigned char *mail_h = nullptr, *mail_d = nullptr;
    size_t mail_used = 0;
    struct MailItem {
        void *host;
        size_t off, bytes;
    };
    std::vector<MailItem> mail_pending;
    bool is_twin = false;
    me_ctx() {
        cloud.p[0] = &own_cloud[0];
        cloud.p[1] = &own_cloud[1];
    }
    me::DevBuf tmp[6];  // scratch
    me::DevBuf red;     // reduction partials
    void *host_pinned = nullptr;
    size_t host_pinned_bytes = 0;
    int shard_rank = 0, shard_world = 1;
    bool borrow_device_input = false;   // me_create flag ME_FLAG_BORROW_DEVICE_INPUT
    bool morton_order = false;          // me_create flag ME_FLAG_MORTON_ORDER: points sorted along the Z curve instead of the Hilbert curve
    me::SlabView slab{-1, 0, 0, 0, 0};  // applied to the next uploads
    bool defer_octree = false;          // index builds leave the octree to cloud_finish_octree (me_run_suite_from)
    double vox_hint = 0;                // > 0: index builds also emit the voxel run records for this voxel size (me_run_suite_from)
    // instrumentation
    bool timers_on = false;
    std::map<std::string, me::TimerRec> timers;
    struct Pending {
        std::string name;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    long long nn_fallback = 0, nn_queries = 0;  // counted only while timers are on
    long long mme_pairs = 0;                     // accepted (query, neighbour) pairs of the MME launches since the last reset (timers on)
    me::DevBuf mme_pairs_buf;
    me::DevBuf mme_refine;                       // k_mme3 -> k_mme_refine: [0] count, [1 ..] sorted offsets of the thin neighbourhoods
    long long mme_refined = 0;                   // queries recomputed by k_mme_refine since the last reset (me_timer_get "mme_refined")
    me::DevBuf nn_far;                           // queries whose octree walk k_nn1 handed over to k_nn_far
    me::DevBuf nn_flags, nn_list_a, nn_list_b;   // 1-NN cascade: unresolved flags of the fine-grid pass, their ordered list, ping-pong
    me::DevBuf mme_keep_e, mme_keep_v;           // me_run_suite_from: the map's per-point MME result across its transform (mme_carry_*)
    long long mme_keep_n = -1;
    me::DevBuf nn1_dbg_buf;                      // octree-walk counters (nodes opened, leaves scanned, points, max per query)
    unsigned long long *nn1_dbg() {
        if (!nn1_dbg_buf.p) {
            if (nn1_dbg_buf.ensure(128) != hipSuccess) return nullptr;
            (void) hipMemsetAsync(nn1_dbg_buf.p, 0, 128, stream);  // ordered before the kernels that count into it
        }
        return nn1_dbg_buf.as<unsigned long long>();
    }

    int fail(int code, const std::string &msg) {
        err = msg;
        return code;
    }
    void shard_range(long long n, long long &b, long long &e) const {
        b = n * shard_rank / shard_world;
        e = n * (shard_rank + 1) / shard_world;
    }
    hipEvent_t get_event();
    void timer_begin(const char *name);
    void timer_end();
    void timers_collect();
};

#define ME_CHECK(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return (ctx)->fail(ME_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__) + " at " + \
                                               __FILE__ + ":" + std::to_string(__LINE__));              \
    } while (0)

#ifdef ME_TRACE  // debug builds: wall-clock marks on stderr (which lane, where) — to locate a stall from outside
#include <chrono>
#define ME_TRACE_POINT(ctx, what)                                                                                              \
    do {                                                                                                                       \
        std::fprintf(stderr, "[me %.3f %s] %s\n",                                                                              \
                     std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(),                \
                     (ctx)->is_twin ? "twin" : "main", what);                                                                   \
        std::fflush(stderr);                                                                                                   \
    } while (0)
#else
#define ME_TRACE_POINT(ctx, what) do { } while (0)
#endif

#define ME_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != ME_OK) return rc__; \
    } while (0)

namespace me {

// RAII scope for a named kernel-family timer
struct TimerScope {  // (scopes do not nest: a scope that calls into another timed phase ends itself first)
    me_ctx *c;
    bool open = true;
    TimerScope(me_ctx *ctx, const char *name) : c(ctx) { c->timer_begin(name); }
    void end() {
        if (open) c->timer_end();
        open = false;
    }
    ~TimerScope() { end(); }
};

// ---- tuning constants.  Compile-time only: the library reads NOTHING from the environment (a drop-in must not change its behaviour
// with the caller's environment).  Measurement builds override them on the command line (make EXTRA=-DME_TUNE_...=v); the values
// below are the measured optima (profiles/EXPERIMENTS.md, profiles/README.md "measurement knobs").
#ifndef ME_TUNE_XCD_CHUNK
#define ME_TUNE_XCD_CHUNK 128     // virtual blocks per XCD chunk (xcd_virtual_block); a huge value = one contiguous piece per XCD
#endif
#ifndef ME_TUNE_SORT_DEPTH
#define ME_TUNE_SORT_DEPTH 2      // Morton levels below the search cell that the sort order (and the choice of the 1-NN grid) may use
#endif
#ifndef ME_TUNE_SORT_PACK
#define ME_TUNE_SORT_PACK 1       // keys-only sort of packed (key, index) words when they fit 64 bits (0: always the pair sort)
#endif
#ifndef ME_TUNE_NN_OCC
#define ME_TUNE_NN_OCC 6.0        // points per occupied cell the 1-NN grid level is chosen for (6 beats 3 and 12 on the bench scene)
#endif
#ifndef ME_TUNE_MME_LEADER_ORIGIN
#define ME_TUNE_MME_LEADER_ORIGIN 1  // k_mme3: moments accumulated about the round leader's point (staged once per candidate) instead of each lane's own query
#endif
#ifndef ME_TUNE_NN_SGPR_MASKS
#define ME_TUNE_NN_SGPR_MASKS 1   // k_nn_grid: compare masks in scalar register pairs (written-out VOP3 encodings) instead of vcc
#endif
#ifndef ME_TUNE_NN1_FAR_CAP
#define ME_TUNE_NN1_FAR_CAP 64    // octree steps after which k_nn1 hands a walk over to k_nn_far (0 = never)
#endif
#ifndef ME_TUNE_NN_FAR_LEAF
#define ME_TUNE_NN_FAR_LEAF 1024  // points a node may hold for k_nn_far to scan it whole instead of descending further
#endif
#ifndef ME_TUNE_MME_WAVES
#define ME_TUNE_MME_WAVES 8  // k_mme3: wavefronts per SIMD the kernel is compiled for (measured 3 / 4 / 6 / 8: profiles/EXPERIMENTS.md "Round 6")
#endif
#ifndef ME_TUNE_MME_REFINE_COND
#define ME_TUNE_MME_REFINE_COND 1.8e-6  // k_mme3 flags a neighbourhood whose smallest covariance eigenvalue is below ~this x cell_h^2 for k_mme_refine (0: never)
#endif
#ifndef ME_TUNE_SUITE_SORT_FIRST
#define ME_TUNE_SUITE_SORT_FIRST 1  // me_run_suite_from: the map's MME is ordered behind the ground truth's radix sort (rocPRIM's onesweep crawls under a full chip)
#endif
#ifndef ME_TUNE_VOX_MERGE_SORT
#define ME_TUNE_VOX_MERGE_SORT 1  // voxel run records sorted by rocPRIM's merge sort (plain kernels) instead of its onesweep radix sort
#endif
#ifndef ME_TUNE_VOX_ONEPASS
#define ME_TUNE_VOX_ONEPASS 1     // voxel tables from ONE pass over the sorted cloud (records about the voxel centres; 0: the three-pass build)
#endif
#ifndef ME_TUNE_SUITE_DEFER_OCTREE
#define ME_TUNE_SUITE_DEFER_OCTREE 1  // me_run_suite_from: the octrees are built after the first MME has been queued, not inside the index build
#endif
#ifndef ME_TUNE_SUITE_VOX_EARLY
#define ME_TUNE_SUITE_VOX_EARLY 1  // me_run_suite_from, second lane: voxel tables whose run records the gather has emitted are finished BEFORE the reverse search (0: after it)
#endif
#ifndef ME_TUNE_SUITE_NN_FIRST
#define ME_TUNE_SUITE_NN_FIRST 1  // me_run_suite_from, second lane: the reverse 1-NN search before the voxel tables (0: round 5's order)
#endif
inline unsigned int xcd_chunk_setting() { return (unsigned int) ME_TUNE_XCD_CHUNK; }

// ---- me_api.hip: copies between caller (host) memory and the device ----
int copy_h2d(me_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);  // ordered on ctx->stream
int copy_d2h(me_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);  // returns when dst holds the data
// ---- me_api.hip: small results to the host without a blit kernel (see me_ctx::mail_h) ----
constexpr size_t kMailBytes = 128 * 1024;
int mail_post(me_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);  // asynchronous on ctx->stream
int mail_sync(me_ctx *ctx);                                                     // hipStreamSynchronize + delivery of what was posted
void mail_drop(me_ctx *ctx);                                                    // forget what was posted (nothing is delivered)
// The host destinations of mail_post are usually LOCALS of the posting function: if it returns before its mail_sync (a failed
// allocation or launch in between), the queued entries must not survive it — the next mail_sync on the context would copy into a
// dead stack frame.  A MailGuard in scope between the first mail_post and the mail_sync drops them on every early return.
struct MailGuard {
    me_ctx *c;
    bool armed = true;
    explicit MailGuard(me_ctx *ctx) : c(ctx) {}
    int sync() {
        armed = false;
        return mail_sync(c);
    }
    ~MailGuard() {
        if (armed) mail_drop(c);
    }
};

// ---- me_prims.hip (rocPRIM-backed primitives) ----
int sort_pairs_u64_u32(me_ctx *ctx, const unsigned long long *k_in, unsigned long long *k_out,
                       const unsigned int *v_in, unsigned int *v_out, long long n, int begin_bit, int end_bit);
int sort_pairs_merge_u64_u32(me_ctx *ctx, const unsigned long long *k_in, unsigned long long *k_out, const unsigned int *v_in,
                             unsigned int *v_out, long long n);
int exclusive_scan_u32(me_ctx *ctx, const unsigned int *in, unsigned int *out, long long n);
int exclusive_scan_u32_plain(me_ctx *ctx, const unsigned int *in, unsigned int *out, long long n);  // no decoupled lookback (in != out)
int cell_start_ranks(me_ctx *ctx, const unsigned long long *codes, long long n, int shift3, unsigned int *out);
int sort_keys_f64(me_ctx *ctx, const double *in, double *out, long long n);
int sort_keys_u64(me_ctx *ctx, const unsigned long long *in, unsigned long long *out, long long n, int begin_bit, int end_bit);
int select_flagged_u32(me_ctx *ctx, const unsigned char *flags, long long n, unsigned int *out, unsigned int *count_device);

// ---- me_index.hip ----
int cloud_upload(me_ctx *ctx, int slot, const double *src, bool src_on_device, long long n, const double *T,
                 double cell_size, bool prefiltered = false);
int cloud_build_index(me_ctx *ctx, int slot, double cell_size);
int cloud_finish_octree(me_ctx *ctx, int slot);
int cloud_finish(me_ctx *ctx, int slot, bool bbox_ready = false);
int cloud_transform(me_ctx *ctx, int slot, const double *T);
int voxel_downsample(me_ctx *ctx, int slot, double voxel_size, long long *n_out);

// ---- me_nn.hip ----
int nn_search(me_ctx *ctx, int qslot, int rslot);
int nn_fetch(me_ctx *ctx, int qslot, int32_t *idx, double *d2);
int nn_unresolved(me_ctx *ctx, int qslot, double *xyz_device, double *d2_device, long long capacity, long long *count);
int nn_points(me_ctx *ctx, int rslot, const double *xyz_device, long long m, double *d2_device, bool bounded, int cov_axis = 0,
              const double *cov_device = nullptr);
int nn_patch(me_ctx *ctx, int qslot, const double *d2_device, long long count);
int nn_cross_message(me_ctx *ctx, double *msg_device, long long cap, long long n_loc_est, long long n_loc_gt, long long counts[2]);
int nn_cross_answer(me_ctx *ctx, const double *gathered_device, int world, long long cap, int own_rank, int dir_mask, int axis,
                    const double *cuts_host, double halo, double *d2_device);
int nn_cross_patch(me_ctx *ctx, const double *d2_reduced_device, long long cap, int own_rank);
int icp_p2p_sums(me_ctx *ctx, int qslot, double max_distance, me_icp_sums *out);
int nn_partial(me_ctx *ctx, int qslot, double gate, int gate_mode, const double trunc[5], me_nn_partial *out);
int nn_sigma(me_ctx *ctx, int qslot, double gate, int gate_mode, const double mean[5], double sigma_num[5]);

// ---- me_reg.hip (registration_methods 1 / 2) ----
int set_normals(me_ctx *ctx, int slot, const double *normals_host);
int get_normals(me_ctx *ctx, int slot, double *normals_host);
int estimate_normals(me_ctx *ctx, int slot, int knn, double *normals_host, int32_t *knn_idx_host, double *knn_d2_host);
int gicp_covariances(me_ctx *ctx, int slot, double epsilon, double *cov_host);
int get_covariances(me_ctx *ctx, int slot, double *cov_host);
int rotate_attributes(me_ctx *ctx, int slot, const double *T);
int icp_lsq_sums(me_ctx *ctx, int qslot, int mode, double max_distance, me_icp_lsq *out);

#ifdef ME_AB  // ---- profiles/ab/me_mme7.hip (round 4's matrix-pipe MME kernel: measurement build only) ----
int mme7_prepare(me_ctx *ctx, Cloud &c, double radius, bool *usable);
int mme7_launch(me_ctx *ctx, Cloud &c, long long b, long long e, unsigned int nb, double radius, int min_k, double *ent_s,
                unsigned char *valid_s, double *part_sum, long long *part_cnt);
#endif

// ---- me_mme.hip ----
int mme_run(me_ctx *ctx, int slot, double radius, int min_k, double *entropies, uint8_t *valid, double *sum_H,
            long long *n_valid);
int mme_fetch(me_ctx *ctx, int slot, double *entropies, uint8_t *valid);
int mme_carry_out(me_ctx *ctx, int slot);  // park the slot's per-point MME result in cloud order (before a transform re-indexes it)
int mme_carry_in(me_ctx *ctx, int slot);   // ... and put it back in the new sorted order

// ---- me_dist.hip (multi-GPU pieces) ----
int halo_pack(me_ctx *ctx, const double *xyz_device, long long n, int axis, const double *cuts_host, int world, double halo,
              double *out_device, long long capacity, long long *counts_host, long long *tags_device = nullptr, long long tag_base = 0);
int voxel_rows_device(me_ctx *ctx, int slot, double voxel_size, double *rows_device, long long capacity, long long *n_rows);
int lattice_messages(me_ctx *ctx, const double *const xyz_device[2], const long long n[2], int clouds, int e0, long long *msg_device);
int lattice_plan(me_ctx *ctx, const long long *msgs_device, int world, int clouds, double halo, int e0, long long *out_host);
int lattice_histograms(me_ctx *ctx, const double *xyz_device, long long n, int e0, int *level, long long origin_bin[3], long long neg_inf[3],
                       unsigned int *hist_device);
int voxel_merge(me_ctx *ctx, int slot, double voxel_size, const double *rows_device, long long m);
int transform_points_device(me_ctx *ctx, double *xyz_device, long long n, const double *T);
int set_mme_result(me_ctx *ctx, int slot, const double *entropies, const uint8_t *valid);
int set_nn_result(me_ctx *ctx, int qslot, int rslot, const double *d2);
int slab_points(me_ctx *ctx, int slot, int64_t *orig_index, uint8_t *owned, long long capacity, long long *count);

// ---- me_render.hip ----
int render_distance(me_ctx *ctx, int qslot, double dis, double gate, int gate_mode, double *rgb, uint8_t *inlier);
int render_entropy(me_ctx *ctx, int slot, double *xyz_out, double *rgb_out, long long capacity, long long *n_valid,
                   double *min_abs_out, double *max_abs_out);

// ---- me_voxel.hip ----
int voxel_build(me_ctx *ctx, int slot, double voxel_size, bool raw);
int voxel_export(me_ctx *ctx, int slot, int32_t *keys, int32_t *npts, double *mu, double *sigma, double *entropy,
                 int64_t *n_voxels);
int awd_scs(me_ctx *ctx, double voxel_size, int min_pts, int scs_radius, double *rows, double *w_sorted,
            int64_t *n_rows, double *awd, double *scs, int64_t counts[3]);
int w2_batch(me_ctx *ctx, const double *mu1, const double *sigma1, const int32_t *n1, const double *mu2,
             const double *sigma2, const int32_t *n2, long long count, double *w);
int scs_table(me_ctx *ctx, const int32_t *keys, const double *w, long long n, int scs_radius, double *scs);

// ---- shared device helpers ----
#ifdef __HIPCC__
// XCD-aware block order.  The hardware deals consecutive block ids round-robin to the 8 XCDs (each with its own L2).
// XCD x gets the chunks x, x+8, x+16, ... of `chunk` consecutive virtual blocks: neighbouring blocks (which stream
// largely the same candidates) share an L2, and every XCD sees a sample of the whole cloud.  One contiguous eighth per
// XCD (the first version) left XCDs with dense regions running long after the others had drained.
// nblocks must be a multiple of 8; the map is a bijection of [0, nblocks).
__device__ __forceinline__ unsigned int xcd_virtual_block(unsigned int bid, unsigned int nblocks, unsigned int chunk) {
    const unsigned int per = nblocks >> 3, x = bid & 7u, s = bid >> 3;
    const unsigned int full = (per / chunk) * chunk;  // slots per XCD covered by whole chunks
    if (s < full) return ((s / chunk) * 8u + x) * chunk + (s % chunk);
    const unsigned int rem = per - full;              // leftover slots: one contiguous piece per XCD
    return full * 8u + x * rem + (s - full);
}

__device__ __forceinline__ double dist2_exact(double ax, double ay, double az, double bx, double by, double bz) {
    // ((dx*dx + dy*dy) + dz*dz) without FMA contraction (file compiled with -ffp-contract=off):
    // this expression must be bit-identical to the CPU path (nanoflann L2 adaptor order).
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0. `sm` holds >= 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double *sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    return r;
}
__device__ __forceinline__ long long block_sum_256_ll(long long v, long long *sm) {
    v = wave_sum_ll(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    long long r = 0;
    if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    return r;
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long x) {
    x &= 0x1fffffULL;
    x = (x | x << 32) & 0x1f00000000ffffULL;
    x = (x | x << 16) & 0x1f0000ff0000ffULL;
    x = (x | x << 8) & 0x100f00f00f00f00fULL;
    x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
    x = (x | x << 2) & 0x1249249249249249ULL;
    return x;
}
__device__ __forceinline__ unsigned int compact21(unsigned long long x) {
    x &= 0x1249249249249249ULL;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ULL;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00fULL;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffULL;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffULL;
    x = (x ^ (x >> 32)) & 0x1fffffULL;
    return (unsigned int) x;
}
__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
    const unsigned int lo = (unsigned int) __builtin_amdgcn_readlane((int) (unsigned int) v, lane);
    const unsigned int hi = (unsigned int) __builtin_amdgcn_readlane((int) (unsigned int) (v >> 32), lane);
    return ((unsigned long long) hi << 32) | lo;
}

// fine (21-bit/axis) grid coordinate of a coordinate in a cloud's frame — the ONE definition every kernel uses
// (division, not reciprocal multiply: (p-o)/(h*2^-s) == ((p-o)/h)*2^s exactly, so a coarser cell is a bit prefix)
__device__ __forceinline__ double fine_coord(double x, double o, double fine_h) { return floor((x - o) / fine_h); }

__device__ __forceinline__ unsigned long long hash_u64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
constexpr unsigned long long kEmptyKey = ~0ULL;

__device__ __forceinline__ int hash_lookup(const unsigned long long *__restrict__ keys,
                                           const unsigned int *__restrict__ vals, unsigned int mask,
                                           unsigned long long key) {
    unsigned int s = (unsigned int) hash_u64(key) & mask;
    for (;;) {
        const unsigned long long k = keys[s];
        if (k == key) return (int) vals[s];
        if (k == kEmptyKey) return -1;
        s = (s + 1) & mask;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Wave-level candidate grouping shared by the radius (MME) and 1-NN grid kernels.
//
// Every lane holds one query and its grid cell (cx,cy,cz).  The group of a round = the pending lanes whose cell is
// within Chebyshev distance 1 of the leader's (first pending lane's) cell.  Their cells span at most 3x3x3, so the
// union of their 3x3x3 neighbourhoods is the group's cell bounding box grown by one: at most 5x5x5 = 125 cells,
// each resolved by ONE hash probe (lane t takes cells t and t+64).  The caller then streams every non-empty run once
// with wave-uniform addresses and lets ALL group lanes test each candidate: compared with one stencil per distinct
// cell this shares candidate fetches between neighbouring cells and keeps the whole wave busy.
// Returns whether this lane is in the group; (rs0,rc0) / (rs1,rc1) = start/count of this lane's two runs.
// ------------------------------------------------------------------------------------------------------------
struct GroupBox {
    int x0, y0, z0, nx, ny;  // origin and x/y extent of the probed cell box (wave-uniform)
};

__device__ __forceinline__ bool wave_group_runs(bool pending, int cx, int cy, int cz, const GridView &g, int cell_lim,
                                                int lane, int &rs0, int &rc0, int &rs1, int &rc1, GroupBox *box = nullptr) {
    const unsigned long long pm = __ballot(pending);  // caller guarantees pm != 0
    const int leader = __ffsll((long long) pm) - 1;
    const int lx = readlane_i(cx, leader), ly = readlane_i(cy, leader), lz = readlane_i(cz, leader);
    const int ex = cx - lx, ey = cy - ly, ez = cz - lz;
    const bool in = pending && ex >= -1 && ex <= 1 && ey >= -1 && ey <= 1 && ez >= -1 && ez <= 1;
    const int x0 = lx - 1 - (__ballot(in && ex < 0) ? 1 : 0), x1 = lx + 1 + (__ballot(in && ex > 0) ? 1 : 0);
    const int y0 = ly - 1 - (__ballot(in && ey < 0) ? 1 : 0), y1 = ly + 1 + (__ballot(in && ey > 0) ? 1 : 0);
    const int z0 = lz - 1 - (__ballot(in && ez < 0) ? 1 : 0), z1 = lz + 1 + (__ballot(in && ez > 0) ? 1 : 0);
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int n_keys = nx * ny * nz;  // <= 125
    if (box) {
        box->x0 = x0;
        box->y0 = y0;
        box->z0 = z0;
        box->nx = nx;
        box->ny = ny;
    }
    rs0 = rc0 = rs1 = rc1 = 0;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int t = lane + 64 * slot;
        if (t < n_keys) {
            const int ix = x0 + t % nx, iy = y0 + (t / nx) % ny, iz = z0 + t / (nx * ny);
            if (ix >= 0 && iy >= 0 && iz >= 0 && ix < cell_lim && iy < cell_lim && iz < cell_lim) {
                const unsigned long long key = spread21((unsigned long long) ix) | (spread21((unsigned long long) iy) << 1) |
                                               (spread21((unsigned long long) iz) << 2);
                const int ci = hash_lookup(g.hkeys, g.hvals, g.hmask, key);
                if (ci >= 0) {
                    const int s = (int) g.cell_start[ci];
                    const int c = (int) g.cell_start[ci + 1] - s;
                    if (slot == 0) {
                        rs0 = s;
                        rc0 = c;
                    } else {
                        rs1 = s;
                        rc1 = c;
                    }
                }
            }
        }
    }
    return in;
}

__device__ __forceinline__ double uniform_f64(double v) {  // a wave-uniform double, moved to a scalar register pair
    const unsigned int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double((int) hi, (int) lo);
}
// Sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48 all end with the total, bit-identical: the two
// additions commute) with gfx950's half-exchange permutes — v_permlane16_swap trades the odd rows of one operand for the
// even rows of the other, v_permlane32_swap the upper half for the lower half — one VALU instruction per 32 bits and
// stage, no LDS round trip.
__device__ __forceinline__ int rows4_sum_i(int v) {
    auto a = __builtin_amdgcn_permlane16_swap((unsigned int) v, (unsigned int) v, false, false);
    v = (int) a[0] + (int) a[1];
    auto b = __builtin_amdgcn_permlane32_swap((unsigned int) v, (unsigned int) v, false, false);
    return (int) b[0] + (int) b[1];
}
__device__ __forceinline__ double rows4_sum_d(double v) {
    unsigned int lo = (unsigned int) __double2loint(v), hi = (unsigned int) __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int) b[0], (int) a[0]) + __hiloint2double((int) b[1], (int) a[1]);
    lo = (unsigned int) __double2loint(v);
    hi = (unsigned int) __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int) d[0], (int) c[0]) + __hiloint2double((int) d[1], (int) c[1]);
}
// ---- cross-lane exchanges on the VALU (DPP) instead of the LDS pipe.  `__shfl_xor` compiles to ds_bpermute_b32: an LDS
// round trip (~100 cycles) per 32 bits and butterfly stage, and the stages depend on each other.  The patterns below are
// plain data-parallel-primitive modifiers of a v_mov: a few cycles each.
// Partner of lane i inside its group of eight, stage 0: i^1 (quad_perm [1,0,3,2]), 1: i^2 (quad_perm [2,3,0,1]),
// 2: 7-i (row_half_mirror: the other quad) — three stages leave a symmetric reduction's result in all eight lanes.
template <int STAGE>
__device__ __forceinline__ int octet_partner_i(int v) {
    constexpr int ctrl = STAGE == 0 ? 0xB1 : (STAGE == 1 ? 0x4E : 0x141);
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, true);
}
template <int STAGE>
__device__ __forceinline__ double octet_partner_d(double v) {
    return __hiloint2double(octet_partner_i<STAGE>(__double2hiint(v)), octet_partner_i<STAGE>(__double2loint(v)));
}
template <int STAGE>
__device__ __forceinline__ long long octet_partner_ll(long long v) {
    const unsigned int lo = (unsigned int) octet_partner_i<STAGE>((int) (unsigned int) v);
    const unsigned int hi = (unsigned int) octet_partner_i<STAGE>((int) (unsigned int) ((unsigned long long) v >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}
// wave-wide min / max of an int, returned wave-uniform (scalar): four DPP stages reduce every row of 16 lanes (i^1, i^2,
// 7-i, row_mirror 15-i), four readlanes and scalar min / max combine the rows
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ------------------------------------------------------------------------------------------------------------
// Wave-level cell-run table for the PER-LANE walkers.  Group = pending lanes whose cell lies within Chebyshev
// distance kGroupR of the leader's; the cell box of the group grown by one (<= 7x7x7 = 343 cells) is resolved with
// one hash probe per (lane, slot) into a wave-private LDS table `tab` (>= kGroupTab entries of {start, count}).
// Every group lane then reads the runs of its own 3x3x3 block from the table: tab[(x-x0) + nx*((y-y0) + ny*(z-z0))].
// 64 curve-consecutive points almost always form one group, so the whole wave walks its candidates together.
// ------------------------------------------------------------------------------------------------------------
constexpr int kGroupR = 2;
constexpr int kGroupTab = 343;                               // (2R+1+2H)^3 for halo H = 1
constexpr int kGroupTab2 = (2 * kGroupR + 5) * (2 * kGroupR + 5) * (2 * kGroupR + 5);  // halo H = 2: 9^3 = 729

// H = halo in cells: 1 when the cell edge covers the search radius, 2 for half-radius cells (5x5x5 stencil).
// CULL: a cell of the box is kept only when it lies within Chebyshev distance H of the cell of SOME group lane (the box is
// the bounding box of the group's cells grown by H: when the cells of 64 curve-consecutive queries do not fill their
// bounding box — an L, a diagonal, a staircase — whole runs of it are adjacent to nobody, and every
// candidate of such a run would be tested by 64 lanes for nothing).  Per (y,z) row of the box the group lanes OR their x
// position into a bit mask (LDS atomics), a row's mask is then dilated over the 3x3 neighbouring rows and by one bit in x:
// a few dozen LDS operations per round against 64 x (9..19) VALU operations per candidate saved.  `rows` = 2 x 49 ints.
// `tcell` (optional) receives the box-relative cell coordinates of every table slot, packed x | y << 3 | z << 6.
// R: Chebyshev radius (in cells) of a group around its leader — kGroupR for the 64-query kernels, 1 for the 16-query
// passes of the MME kernel (a (2R+1+2H)^3 = 125-entry table).
// `tabc` / `cell_aux` (optional, both or neither): tabc[slot] = cell_aux[cell index] for every non-empty slot — a second per-cell
// array looked up with the same probe (the matrix-pipe MME kernel keeps the first feature chunk of every cell there).
template <int H = 1, bool CULL = false, int R = kGroupR, bool MLP = false>
__device__ __forceinline__ bool wave_group_table(bool pending, int cx, int cy, int cz, const GridView &g, int cell_lim,
                                                 int lane, int2 *tab, GroupBox &box, int *n_keys_out = nullptr,
                                                 unsigned int *rows = nullptr, unsigned short *tcell = nullptr,
                                                 unsigned int *tabc = nullptr, const unsigned int *cell_aux = nullptr,
                                                 int *leader_out = nullptr) {
    const unsigned long long pm = __ballot(pending);  // caller guarantees pm != 0
#ifdef ME_LEADER_FIRST
    const int leader = __ffsll((long long) pm) - 1;
#else
    // The leader is the MIDDLE pending lane, not the first: the lanes hold curve-consecutive points, the first pending lane
    // sits at one end of the stretch of space they cover and its Chebyshev ball reaches half as far into it.
    const int rank = __popcll(pm & ((1ULL << lane) - 1ULL));
    const int n_pending = __popcll(pm);
    int leader = __ffsll((long long) __ballot(pending && rank == (n_pending >> 1))) - 1;
#ifndef ME_LEADER_MIDDLE
    // Round 4: when the middle lane's group does not hold every pending lane, the lanes at the quartiles of the pending stretch
    // are tried as well — ranks k/6 of it — and the leader whose group is the largest wins (a few ballots; a round costs thousands of
    // instructions).
    {
        auto group_size = [&](int l) {
            const int ax = readlane_i(cx, l), ay = readlane_i(cy, l), az = readlane_i(cz, l);
            const int ux = cx - ax, uy = cy - ay, uz = cz - az;
            return __popcll(__ballot(pending && ux >= -R && ux <= R && uy >= -R && uy <= R && uz >= -R && uz <= R));
        };
        int best = group_size(leader);
        if (best < n_pending && n_pending >= 8) {
#pragma unroll
            for (int c = 1; c <= 5; ++c) {
                if (c == 3) continue;  // (the middle: tried already)
                const int l = __ffsll((long long) __ballot(pending && rank == ((c * n_pending) / 6))) - 1;
                const int b = group_size(l);
                if (b > best) {
                    best = b;
                    leader = l;
                }
            }
        }
    }
#endif
#endif
    if (leader_out) *leader_out = leader;
    const int lx = readlane_i(cx, leader), ly = readlane_i(cy, leader), lz = readlane_i(cz, leader);
    const int ex = cx - lx, ey = cy - ly, ez = cz - lz;
    const bool in = pending && ex >= -R && ex <= R && ey >= -R && ey <= R && ez >= -R && ez <= R;
    // (readfirstlane: the butterfly leaves the same value in every lane, but only this tells the compiler so — without it the
    // box, the table size and everything the callers derive from them live in vector registers and loops over them diverge)
    const int x0 = __builtin_amdgcn_readfirstlane(wave_min_i(in ? cx : lx)) - H;
    const int x1 = __builtin_amdgcn_readfirstlane(wave_max_i(in ? cx : lx)) + H;
    const int y0 = __builtin_amdgcn_readfirstlane(wave_min_i(in ? cy : ly)) - H;
    const int y1 = __builtin_amdgcn_readfirstlane(wave_max_i(in ? cy : ly)) + H;
    const int z0 = __builtin_amdgcn_readfirstlane(wave_min_i(in ? cz : lz)) - H;
    const int z1 = __builtin_amdgcn_readfirstlane(wave_max_i(in ? cz : lz)) + H;
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int n_keys = nx * ny * nz;  // <= (2R+1+2H)^3
    if (n_keys_out) *n_keys_out = n_keys;
    box.x0 = x0;
    box.y0 = y0;
    box.z0 = z0;
    box.nx = nx;
    box.ny = ny;
    if (CULL) {
        const int n_rows = ny * nz;  // <= 49 (H = 1)
        if (lane < n_rows) rows[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (in) atomicOr(&rows[(cy - y0) + ny * (cz - z0)], 1u << (cx - x0));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < n_rows) {
            const int ry = lane % ny, rz = lane / ny;
            unsigned int m = 0;
            for (int dz = -H; dz <= H; ++dz)
                for (int dy = -H; dy <= H; ++dy) {
                    const int yy = ry + dy, zz = rz + dz;
                    if (yy >= 0 && yy < ny && zz >= 0 && zz < nz) m |= rows[yy + ny * zz];
                }
            unsigned int d = m;
#pragma unroll
            for (int s = 1; s <= H; ++s) d |= (m << s) | (m >> s);
            rows[64 + lane] = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (MLP) {
        // Memory-level parallelism for kernels that run few waves per SIMD: the probes of ALL table slots of a lane are issued
        // before any is waited for (the plain loop below is four dependent round trips per 64 slots, one batch after the other).
        constexpr int NIT = ((2 * R + 1 + 2 * H) * (2 * R + 1 + 2 * H) * (2 * R + 1 + 2 * H) + 63) / 64;
        unsigned long long key[NIT], got[NIT];
        unsigned int slot[NIT];
        bool want[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int t = lane + 64 * i;
            const int tx = t % nx, ty = (t / nx) % ny, tz = t / (nx * ny);
            const int ix = x0 + tx, iy = y0 + ty, iz = z0 + tz;
            want[i] = t < n_keys && ix >= 0 && iy >= 0 && iz >= 0 && ix < cell_lim && iy < cell_lim && iz < cell_lim;
            if (CULL) want[i] = want[i] && ((rows[64 + (t < n_keys ? ty + ny * tz : 0)] >> tx) & 1u);
            key[i] = spread21((unsigned long long) ix) | (spread21((unsigned long long) iy) << 1) | (spread21((unsigned long long) iz) << 2);
            slot[i] = (unsigned int) hash_u64(key[i]) & g.hmask;
            got[i] = kEmptyKey;
            if (tcell && t < n_keys) tcell[t] = (unsigned short) (tx | (ty << 3) | (tz << 6));
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (want[i]) got[i] = g.hkeys[slot[i]];
        int ci[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            // (collisions: linear probing, rare)
            while (want[i] && got[i] != key[i] && got[i] != kEmptyKey) {
                slot[i] = (slot[i] + 1) & g.hmask;
                got[i] = g.hkeys[slot[i]];
            }
            want[i] = want[i] && got[i] == key[i];
            ci[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (want[i]) ci[i] = (int) g.hvals[slot[i]];
        unsigned int cs0[NIT], cs1[NIT], ax[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            cs0[i] = cs1[i] = ax[i] = 0;
            if (want[i]) {
                cs0[i] = g.cell_start[ci[i]];
                cs1[i] = g.cell_start[ci[i] + 1];
                if (tabc) ax[i] = cell_aux[ci[i]];
            }
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int t = lane + 64 * i;
            if (t < n_keys) {
                tab[t] = make_int2((int) cs0[i], (int) (cs1[i] - cs0[i]));
                if (tabc) tabc[t] = ax[i];
            }
        }
    } else
    for (int t = lane; t < n_keys; t += 64) {
        const int tx = t % nx, ty = (t / nx) % ny, tz = t / (nx * ny);
        const int ix = x0 + tx, iy = y0 + ty, iz = z0 + tz;
        int2 run = make_int2(0, 0);
        unsigned int aux = 0;
        bool want = ix >= 0 && iy >= 0 && iz >= 0 && ix < cell_lim && iy < cell_lim && iz < cell_lim;
        if (CULL) want = want && ((rows[64 + ty + ny * tz] >> tx) & 1u);
        if (want) {
            const unsigned long long key = spread21((unsigned long long) ix) | (spread21((unsigned long long) iy) << 1) |
                                           (spread21((unsigned long long) iz) << 2);
            const int ci = hash_lookup(g.hkeys, g.hvals, g.hmask, key);
            if (ci >= 0) {
                run.x = (int) g.cell_start[ci];
                run.y = (int) g.cell_start[ci + 1] - run.x;
                if (tabc) aux = cell_aux[ci];
            }
        }
        tab[t] = run;
        if (tabc) tabc[t] = aux;
        if (tcell) tcell[t] = (unsigned short) (tx | (ty << 3) | (tz << 6));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return in;
}
constexpr int kGroupRows = 128;  // LDS ints per wave for the CULL row masks (64 raw + 64 dilated)

// Calls f(begin, end, slot) once per non-empty run of the wave's table, with WAVE-UNIFORM arguments (so that a loop over
// [begin, end) fetches candidates with scalar loads and all lanes test the same candidate); slot = its table index.
// MERGE: consecutive table slots whose runs are contiguous in the sorted array (the curve visits cell x + 1 right after
// cell x) are handed over as ONE run — with 6-point cells the per-run set-up of the 1-NN grid kernel (staging a tile,
// two barriers, the tail of the group-of-four loop) costs more than ranking the run's candidates.
template <bool MERGE = false, class F>
__device__ __forceinline__ void wave_for_each_run(const int2 *tab, int n_keys, int lane, F &&f) {
    for (int base = 0; base < n_keys; base += 64) {
        const int t = base + lane;
        const int cnt = (t < n_keys) ? tab[t].y : 0;
        unsigned long long m = __ballot(cnt > 0);
        while (m) {
            const int n = __ffsll((long long) m) - 1;
            m &= m - 1;
            const int2 run = tab[base + n];
            const int cs = __builtin_amdgcn_readfirstlane(run.x);
            int cc = __builtin_amdgcn_readfirstlane(run.y);
            if (MERGE) {
                while (m) {
                    const int n2 = __ffsll((long long) m) - 1;
                    const int2 nx = tab[base + n2];
                    if (__builtin_amdgcn_readfirstlane(nx.x) != cs + cc) break;
                    cc += __builtin_amdgcn_readfirstlane(nx.y);
                    m &= m - 1;
                }
            }
            f(cs, cs + cc, base + n);
        }
    }
}
#endif

}  // namespace me
