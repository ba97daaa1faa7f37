/*
 * mapeval_hip.h — C ABI of libmapeval_hip.so: MapEval's metric hot path on AMD MI355X (gfx950).
 *
 * The reference (JokerJohn/Cloud_Map_Evaluation) has no plugin / FFI layer: its seam is the set of MapEval member
 * calls made by MapEval::process() (map_eval/src/map_eval.cpp:52-85).  Each entry point below replaces one of
 * those calls (or the Open3D / VoxelCalculator operator it loops over) with ONE batched device call; the
 * reference file:line it stands in for is cited per function.  Plain pointers and sizes only — no C++ or torch
 * types cross this boundary.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - clouds are AoS fp64 `double[N][3]`, exactly open3d::geometry::PointCloud::points_.data() (map_eval.h:45);
 *     host pointers unless the function name ends in _device.
 *   - the caller owns every host buffer; the library owns all device memory inside me_ctx.
 *   - every call is synchronous on return and returns ME_OK (0) or a negative ME_ERR_*; me_last_error() has text.
 *   - nullable outputs may be NULL (skips the D2H copy).
 *   - one me_ctx = one GPU = one host thread at a time (MapEval::process is single-threaded, map_eval.cpp:4).
 *   - results follow the reference's arithmetic, quirks included (squared-vs-unsquared gate :1219, triple
 *     division of sigma voxel_calculator.cpp:48/102/120, Cholesky-trace "W2" :136-138, k>=10 / k>=5 MME gates).
 *     Inlier / valid / voxel point counts are bit-exact; floating-point sums agree to ~1e-12 relative
 *     (summation order differs, as it already does between two runs of the TBB/OpenMP reference).
 */
#ifndef MAPEVAL_HIP_H
#define MAPEVAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_OK 0
#define ME_ERR_ARG (-1)      /* bad argument (null pointer, bad slot, n out of range, ...)          */
#define ME_ERR_HIP (-2)      /* a HIP runtime call failed (no device, OOM, ...)                     */
#define ME_ERR_STATE (-3)    /* call order violated (cloud not uploaded, search not run, ...)       */
#define ME_ERR_CAPACITY (-4) /* caller-provided output capacity too small (count still returned)    */

#define ME_SLOT_EST 0 /* map_3d_  (map_eval.h:322) */
#define ME_SLOT_GT 1  /* gt_3d_   (map_eval.h:322) */

#define ME_GATE_LE_UNSQUARED 0 /* keep iff d2 <= gate        — calculateMetricsWithInitialMatrix, map_eval.cpp:1219 (sic) */
#define ME_GATE_LT_SQUARED 1   /* keep iff d2 <  gate*gate   — Open3D EvaluateRegistration / ICP, map_eval.cpp:1168        */

typedef struct me_ctx me_ctx;

/* Raw (shard-local, all-reduce-able) sums of one direction of the AC/COM/CD pass.
 * Every field is a plain sum over the query points this context owns, so partials from several GPUs add. */
typedef struct me_nn_partial {
    int64_t n_query;     /* queries processed (shard size)                                  */
    int64_t n_corr;      /* correspondences that passed the gate  (points_set.size(), :1076) */
    int64_t n_inl[5];    /* number_vec   (map_eval.cpp:1102,1107,...)                        */
    double sum_d[5];     /* mean_vec before /C  (:1100)                                      */
    double sum_d2[5];    /* rmse_vec before /C  (:1101)                                      */
    double sum_sqrt_all; /* sum of sqrt(d2) over ALL queries, ungated (computeChamferDistance :1416) */
} me_nn_partial;

/* Finished result block of getDiffRegResultWithCorrespondence (map_eval.cpp:1140-1144):
 * the five Vector5d pushed into est_gt_results / gt_est_results, plus the CD term. */
typedef struct me_nn_stats_out {
    int64_t n_src;       /* source.points_.size() (whole cloud)                              */
    int64_t n_corr;      /* C                                                                */
    double mean[5];      /* result[0] */
    double rmse[5];      /* result[1]  -> "RMSE/AC:"  (map_eval.cpp:439)                     */
    double fitness[5];   /* result[2]  -> "Comp:"     (map_eval.cpp:445)                     */
    double sigma[5];     /* result[3] */
    double number[5];    /* result[4] */
    double mean_nn_dist; /* sum_sqrt_all / n_src : one half of computeChamferDistance (:1429) */
} me_nn_stats_out;

/* ---- lifetime ---------------------------------------------------------------------------------------------- */
/* device: HIP device ordinal (one context per GPU).  flags: 0, or an OR of the ME_FLAG_* below.  NULL on failure.
 * The library reads nothing from the environment: every behaviour switch is a flag here or a compile-time constant. */
me_ctx *me_create(int device, int flags);
/* ME_FLAG_BORROW_DEVICE_INPUT: me_upload_cloud_device / me_upload_slab_device without a transform (T NULL or the identity) read the
 * caller's device buffer WHERE IT LIES instead of copying it: the caller keeps it valid and unchanged until the slot's next upload
 * (or me_destroy).  Saves one 48-byte-per-point pass per cloud; calls that modify the cloud in place (me_transform_cloud,
 * me_voxel_downsample) switch to a private copy first.  No reference counterpart (Open3D owns its points_). */
#define ME_FLAG_BORROW_DEVICE_INPUT 1
/* ME_FLAG_MORTON_ORDER: sort the points along the Z (Morton) curve instead of the Hilbert curve.  An implementation detail of the index
 * — counts identical, fp results equal to rounding (tests/test_gpu_order.py) — kept as a test / measurement switch. */
#define ME_FLAG_MORTON_ORDER 2
void me_destroy(me_ctx *ctx);
const char *me_last_error(me_ctx *ctx); /* ctx may be NULL: returns the last me_create error */
int me_version(void);

/* A second LANE on the same clouds: a context that shares both cloud slots with `ctx` (uploads, indexes, NN / MME / voxel
 * results are common) but owns its stream, scratch memory and timers.  Two host threads may then drive the two
 * contexts concurrently, e.g. indexing the ground truth (HBM-bound) under the MME pass of the map (VALU-bound).
 * Rules: never let both lanes upload / re-index / transform the same slot, or write the same product of a slot (its
 * NN result, MME, voxel table), at the same time; reading a slot's index while the other lane builds the OTHER slot is
 * fine.  The twin is owned by `ctx` (me_destroy(ctx) frees it; me_destroy(twin) is a no-op).  No reference counterpart. */
me_ctx *me_twin(me_ctx *ctx);

/* Multi-GPU slab sharding (no reference counterpart; the reference is single-process).  After this call every
 * per-point pass (NN, MME) only processes the `rank`-th of `world` equal slabs of the sorted (space-filling-curve) query order,
 * and the voxel passes only own voxels whose key index falls in the rank's slab; partial sums are returned for
 * the caller to all-reduce (RCCL).  Default (0,1) = whole job. */
int me_set_shard(me_ctx *ctx, int rank, int world);

/* Multi-GPU SPATIAL slab (no reference counterpart).  After this call me_upload_cloud* keeps only the points with
 * lo - halo <= p[axis] < hi + halo (after the transform); points with lo <= p[axis] < hi are OWNED by this context —
 * they are the queries of every per-point pass and the only contributors to the voxel partials — the rest are halo,
 * visible as reference points / neighbours only.  Every rank therefore sorts and indexes ~1/world of each cloud.
 * MME is exact when halo >= nn_radius.  1-NN is exact for every query whose best distance is below its distance to
 * the slab's outer faces; the others are returned by me_nn_unresolved for the cross-rank step (me_nn_points on every
 * rank, min-reduce, me_nn_patch).
 * Per-point outputs in slab mode (round 3; what the distributed host needs for map_entropy.pcd / raw_rendered_dis_map.pcd,
 * map_eval.cpp:485-495, 686-736): the entropies / valid arrays of me_mme and the idx / d2 arrays of me_nn1 / me_nn_fetch have
 * one entry per point this context HOLDS of the slot (me_cloud_size: owned + halo), in the order me_slab_points reports;
 * halo points read entropy 0 / valid 0 / d2 -1 / idx -1; idx is an index into the points held of the reference slot and is
 * the global neighbour only where the query did not go through the cross-rank step (d2 is global after me_nn_patch).
 * axis < 0 switches slab mode off.  Must be called before the uploads it applies to. */
int me_set_slab(me_ctx *ctx, int axis, double lo, double hi, double halo);

/* Slab mode, 1-NN cross-rank step.  me_nn_unresolved: the owned queries of the last me_nn1(query_slot, ..) whose
 * result is not yet provably global; xyz_device (capacity x 3, device memory) receives their coordinates, *count
 * their number (ME_ERR_CAPACITY if it exceeds capacity; xyz_device may be NULL to query the count); d2_device
 * (optional, capacity doubles) receives their current best squared distances, the bound the other ranks have to beat. */
int me_nn_unresolved(me_ctx *ctx, int query_slot, double *xyz_device, double *d2_device, int64_t capacity, int64_t *count);
/* Exact squared distance from each of m arbitrary points (device, m x 3) to the nearest point this context holds of
 * ref_slot (owned + halo) -> d2_device[m]. */
int me_nn_points(me_ctx *ctx, int ref_slot, const double *xyz_device, int64_t m, double *d2_device);
/* Same with an upper bound per point: on entry d2_inout_device[i] bounds the answer (+inf = none), on exit it holds
 * min(bound, nearest squared distance here).  A rank whose points are all farther prunes at the root, which is what
 * makes the cross-rank step cheap: the querying rank passes the distance it already has.  A NEGATIVE bound marks a slot that
 * needs no answer (the padding of a fixed-capacity message, a rank's own queries): it comes back unchanged after one step. */
int me_nn_points_bounded(me_ctx *ctx, int ref_slot, const double *xyz_device, int64_t m, double *d2_inout_device);
/* me_nn_points_bounded for queries whose OWNER has already searched everything in a band of one axis: covered_device[2 i],
 * covered_device[2 i + 1] = [lo, hi) along `axis` — the owner's slab + halo, which holds every point of the cloud in that band
 * (me_halo_pack_device) — and d2_inout_device[i] = the nearest squared distance found there.  The caller thereby guarantees that no
 * point inside the band is closer than the bound; this rank then only has to look at the part of its tree OUTSIDE the band.  Same
 * result as me_nn_points_bounded — min(bound, nearest squared distance here) — for a fraction of the walk: without it a neighbour
 * disproves a far outlier (a ball of metres reaching across the face) cell by cell inside the strip both ranks hold
 * (the computeChamferDistance / getDiffRegResult searches have no distance limit, map_eval.cpp:1398-1431, :1100-1110). */
int me_nn_points_covered(me_ctx *ctx, int ref_slot, const double *xyz_device, int64_t m, double *d2_inout_device, int axis,
                         const double *covered_device);
/* The cross-rank step above as three calls on ONE fixed-capacity message per rank (round 5) — what the ranks all-gather and min-reduce:
 *   row 0                          [open queries map -> gt, open queries gt -> map, n_local_est, n_local_gt]
 *   rows 1 .. capacity             the open queries of the last me_nn1(ME_SLOT_EST, ME_SLOT_GT): x, y, z, bound (best squared distance so far)
 *   rows 1 + capacity .. 2 capacity  the same for me_nn1(ME_SLOT_GT, ME_SLOT_EST); unused rows carry the bound -1
 * me_nn_cross_message  writes this rank's message (1 + 2 capacity rows x 4 doubles, device memory); counts[2] = its open queries per
 *                      direction (more than `capacity`: the message carries the first `capacity`, the caller falls back to exactly
 *                      sized messages through me_nn_unresolved / me_nn_points_covered / me_nn_patch).
 * me_nn_cross_answer   the all-gathered messages of all ranks (world x (1 + 2 capacity) x 4) -> d2_device (world x (1 + 2 capacity)):
 *                      every slot of another rank = min(its bound, the nearest squared distance among the points held here, looking only
 *                      OUTSIDE the band [cuts[k] - halo, cuts[k + 1] + halo) of `axis` its owner k has searched completely — as
 *                      me_nn_points_covered); own slots and padding keep their bound, every rank's header slot is written as 0.  dir_mask: bit 0 / 1 = answer the map -> gt /
 *                      gt -> map block (a direction nobody else has open queries in is copied through).  cuts: host, world + 1 values.
 * me_nn_cross_patch    after the all-reduce MIN of d2_device over the ranks: this rank's block patches its open queries (me_nn_patch for
 *                      both directions at once). */
int me_nn_cross_message(me_ctx *ctx, double *msg_device, int64_t capacity, int64_t n_local_est, int64_t n_local_gt, int64_t counts[2]);
int me_nn_cross_answer(me_ctx *ctx, const double *gathered_device, int world, int64_t capacity, int own_rank, int dir_mask, int axis,
                       const double *cuts, double halo, double *d2_device);
int me_nn_cross_patch(me_ctx *ctx, const double *d2_reduced_device, int64_t capacity, int own_rank);
/* Overwrites the squared distances of the unresolved queries (same order as me_nn_unresolved returned them) with the
 * globally min-reduced values d2_device[count]. */
int me_nn_patch(me_ctx *ctx, int query_slot, const double *d2_device, int64_t count);
/* The per-point result of the last me_nn1(query_slot, ..) as it stands now (after me_nn_patch in slab mode): idx / d2 as
 * me_nn1 returns them (either may be NULL). */
int me_nn_fetch(me_ctx *ctx, int query_slot, int32_t *idx, double *d2);
/* Slab mode: the points this context holds of `slot`, in the order of its per-point outputs: orig_index[i] = the point's
 * position in the array that was uploaded (identity after me_upload_slab_device), owned[i] = 1 for the slab's own points,
 * 0 for halo.  Either array may be NULL; *count = me_cloud_size.  ME_ERR_CAPACITY when capacity < count. */
int me_slab_points(me_ctx *ctx, int slot, int64_t *orig_index, uint8_t *owned, int64_t capacity, int64_t *count);
/* Hand a context per-point results that were computed elsewhere (by the ranks of a distributed run, put together with
 * me_slab_points): afterwards me_render_entropy(slot) / me_render_distance(query_slot, ..) colour the WHOLE cloud held by
 * this context exactly as after me_mme / me_nn1 (map_entropy.pcd, raw_rendered_dis_map.pcd; map_eval.cpp:485-495, 686-736).
 * Arrays are host memory in cloud order, one entry per point of the slot. */
int me_set_mme_result(me_ctx *ctx, int slot, const double *entropies, const uint8_t *valid);
int me_set_nn_result(me_ctx *ctx, int query_slot, int ref_slot, const double *d2);

/* Slab mode, voxel partials: Gaussians of the OWNED points only, RAW second moments (M2 = sum (p-mu)(p-mu)^T, no
 * division), ascending key order; partials of the same voxel from different ranks merge with Chan's formula. */
int me_voxel_partials(me_ctx *ctx, int slot, double voxel_size, int32_t *keys /*V x 3*/, int32_t *npts /*V*/,
                      double *mu /*V x 3*/, double *m2 /*V x 9*/, int64_t *n_voxels);

/* Multi-GPU with DISTRIBUTED INPUT (no reference counterpart): every rank starts with 1/world of each cloud.
 *   me_transform_points_device   *cloud = cloud->Transform(T) (map_eval.cpp:1206) on a raw device buffer, in place — applied to
 *                                a rank's part of the estimated map BEFORE the exchange, so that slab membership is decided
 *                                on the exact transformed coordinate.
 *   me_halo_pack_device          the send side of the one-shot halo exchange.  cuts[world + 1] (host, ascending, cuts[0] = -inf,
 *                                cuts[world] = +inf) are the slab faces along `axis`; point p goes to EVERY rank k with
 *                                cuts[k] - halo <= p[axis] < cuts[k+1] + halo (the filter me_set_slab applies on the
 *                                receiving side).  out_device (capacity x 3) receives the points destination-major, in input
 *                                order inside a destination (deterministic); counts[world] the segment sizes — exactly the
 *                                send buffer and split sizes of one all_to_all.  out_device == NULL: counts only.
 *   me_voxel_partial_rows_device me_voxel_partials as rows [kx, ky, kz, n, mu(3), M2(9)] (16 doubles) in a device buffer:
 *                                what the all-gather of the voxel partials carries (rows with n == 0 are padding).
 *   me_voxel_merge_device        Chan's parallel update of the gathered rows of all ranks -> the slot's voxel table exactly as
 *                                VoxelCalculator::buildVoxelMap leaves it (voxel_calculator.cpp:21-56: M2/(n-1)^2 for n > 10);
 *                                when both slots hold a merged table of the same voxel size, me_awd_scs runs on them.
 *   me_upload_slab_device        me_upload_cloud_device for points that ARE this rank's slab + halo already (what the exchange
 *                                delivered, transform applied): me_set_slab's filter pass is skipped, ownership still follows
 *                                the slab. */
int me_transform_points_device(me_ctx *ctx, double *xyz_device, int64_t n, const double *T_rowmajor4x4);
int me_upload_slab_device(me_ctx *ctx, int slot, const double *xyz_device, int64_t n, double cell_size);
int me_halo_pack_device(me_ctx *ctx, const double *xyz_device, int64_t n, int axis, const double *cuts, int world, double halo,
                        double *out_device, int64_t capacity, int64_t *counts);
/* me_halo_pack_device that also packs, in the same order, tag_base + the input index of every copied point into tags_device
 * (capacity entries; may be NULL): the receiver of the exchange then knows which point of the WHOLE cloud an entry of its slab is
 * (the C++ host's per-point outputs: map_entropy.pcd, raw_rendered_dis_map.pcd, map_eval.cpp:485-495, 686-736). */
int me_halo_pack_tagged_device(me_ctx *ctx, const double *xyz_device, int64_t n, int axis, const double *cuts, int world, double halo,
                               double *out_device, int64_t *tags_device, int64_t tag_base, int64_t capacity, int64_t *counts);
/* voxel_size > 0: every index build of this context (and of its twin) from now on also emits the voxel run records of
 * VoxelCalculator::buildVoxelMap for that voxel size (voxel_calculator.cpp:21-56) while it gathers the sorted cloud; a later
 * me_voxel_gaussians / me_voxel_partials / me_voxel_partial_rows_device / me_awd_scs with the SAME size then has no pass over the cloud
 * left (a sort and a reduction of ~n / 60 records).  Results as without the hint (keys and populations exact, sums to 1e-9 of their
 * scale).  0: off (the default).  me_run_suite_from does this by itself for the duration of the call. */
int me_set_voxel_hint(me_ctx *ctx, double voxel_size);
/* The lean exchange (no reference counterpart; dist.py `lattice_plan`): the marginal histograms of a rank's part of a cloud on an
 * ABSOLUTE power-of-two lattice.  Bin i of axis a counts the finite coordinates with floor(x_a / w) == origin_bin[a] + i,
 * w = 2^(e0 + *level); *level is the smallest one for which every axis of this buffer fits ME_LATTICE_BINS bins.  neg_inf[a]
 * counts the coordinates that are -inf (me_halo_pack_device hands those to rank 0; NaN and +inf satisfy no slab's test).
 * hist_device: 3 x ME_LATTICE_BINS uint32, axis-major.  With the histograms of every rank's parts (one all-gather) each rank computes
 * the slab cuts AT BIN EDGES, a halo of whole bins and the exact size of every message of the halo exchange — x / w, floor and
 * (whole number) * w are exact in fp64, so me_halo_pack_device's comparisons against such cuts decide exactly what the bins say. */
#define ME_LATTICE_BINS 4096
int me_lattice_histograms_device(me_ctx *ctx, const double *xyz_device, int64_t n, int e0, int32_t *level, int64_t origin_bin[3],
                                 int64_t neg_inf[3], uint32_t *hist_device);
/* me_lattice_histograms_device for a rank's `clouds` (1 or 2) pieces at once, written as the rows of its gather message:
 * msg_device = clouds x (8 + 3 ME_LATTICE_BINS) int64, row = [level, origin_bin x y z, n, neg_inf x y z | the counts, axis-major] —
 * what me_lattice_plan_device reads.  Two launches per piece, one host read and one stream synchronisation for both. */
int me_lattice_messages_device(me_ctx *ctx, const double *xyz_a_device, int64_t n_a, const double *xyz_b_device, int64_t n_b, int clouds, int e0,
                               int64_t *msg_device);
/* The plan of the lean exchange from the gathered messages of all ranks: msgs_device = world x clouds rows (rank-major; the LAST cloud is
 * the ground truth, whose extent picks the slab axis) of 8 + 3 ME_LATTICE_BINS int64: [level, origin_bin x y z, n, neg_inf x y z |
 * me_lattice_histograms_device's counts, axis-major].  out (host, 4 + clouds + world - 1 + world * clouds * world int64):
 * [axis, level of the combined window, its first bin on that axis, halo in bins, points per cloud, the cut edges c_1 .. c_{world-1}
 * (bins from the window's first), then counts[source rank][cloud][destination rank]] — cut k = (first bin + c_k) * 2^(e0 + level),
 * halo = (halo in bins) * 2^(e0 + level).  Four small kernels; the same arithmetic, number for number, as dist.lattice_plan. */
int me_lattice_plan_device(me_ctx *ctx, const int64_t *msgs_device, int world, int clouds, double halo, int e0, int64_t *out);
int me_voxel_partial_rows_device(me_ctx *ctx, int slot, double voxel_size, double *rows_device, int64_t capacity, int64_t *n_rows);
int me_voxel_merge_device(me_ctx *ctx, int slot, double voxel_size, const double *rows_device, int64_t n_rows);

/* ---- clouds ------------------------------------------------------------------------------------------------ */
/* Replaces: *map_3d_ = map_3d_->Transform(initial_matrix) (map_eval.cpp:1206) + every KDTreeFlann::SetGeometry
 * (map_eval.cpp:1214,1227,1401-1402,1449,1551,1619): uploads the cloud, applies T (row-major 4x4, NULL = none;
 * homogeneous divide as Open3D), sorts it along a space-filling curve and builds the search index ONCE.
 * cell_size: edge of the radius-search grid cell (pass nn_radius; <= 0 = automatic, rebuilt lazily by me_mme). */
int me_upload_cloud(me_ctx *ctx, int slot, const double *xyz_host, int64_t n, const double *T_rowmajor4x4,
                    double cell_size);
int me_upload_cloud_device(me_ctx *ctx, int slot, const double *xyz_device, int64_t n,
                           const double *T_rowmajor4x4, double cell_size);
/* open3d::geometry::PointCloud::VoxelDownSample (map_eval.cpp:38-39) on the cloud already on the device, in place:
 * voxel index = floor((p - (min_bound - voxel_size/2)) / voxel_size), one output point per occupied voxel = the mean of
 * its points accumulated in cloud order (bit-identical to the CPU arithmetic).  Output order: ascending voxel index
 * (Open3D: hash-map iteration order).  The index is rebuilt; *n_out = points kept.   */
int me_voxel_downsample(me_ctx *ctx, int slot, double voxel_size, int64_t *n_out);
/* *cloud = cloud->Transform(T) (map_eval.cpp:1206, :1392) on the cloud already on the device (row-major 4x4); the index
 * is rebuilt.  Lets MME run on the map as loaded and AC/COM/CD/AWD on the transformed map, as the reference does. */
int me_transform_cloud(me_ctx *ctx, int slot, const double *T_rowmajor4x4);
int64_t me_cloud_size(me_ctx *ctx, int slot);
/* transformed points back to the host (N x 3), original order — what map_3d_->points_ holds after :1206 */
int me_download_cloud(me_ctx *ctx, int slot, double *xyz_host);

/* ---- 1-NN: KDTreeFlann::SearchKNN(q, 1, idx, d2) over a whole cloud (map_eval.cpp:1218,1231,1415,1424,579) --- */
/* Searches every point of query_slot in ref_slot; results stay on the device for the me_nn_* calls below.
 * idx / d2 (query-cloud order, length N_query; nullable) receive the neighbour index in ref cloud order and the
 * SQUARED distance ((dx*dx + dy*dy) + dz*dz), bit-identical to the CPU path.  Ties -> smallest ref index. */
int me_nn1(me_ctx *ctx, int query_slot, int ref_slot, int32_t *idx, double *d2);

/* getDiffRegResultWithCorrespondence / getDiffRegResult (map_eval.cpp:1069-1145, 828-897, 990-1067) on the
 * correspondences of the last me_nn1(query_slot, ...): gate (negative = none) + 5 thresholds.  One-shot,
 * single GPU. */
int me_nn_stats(me_ctx *ctx, int query_slot, double gate, int gate_mode, const double trunc[5],
                me_nn_stats_out *out);
/* The same, split for multi-GPU: raw partial sums -> (all-reduce) -> second pass for sigma -> finalize. */
int me_nn_partial_sums(me_ctx *ctx, int query_slot, double gate, int gate_mode, const double trunc[5],
                       me_nn_partial *out);
int me_nn_sigma_sums(me_ctx *ctx, int query_slot, double gate, int gate_mode, const double mean[5],
                     double sigma_num[5]);
void me_nn_finalize(const me_nn_partial *total, const double sigma_num[5], int64_t n_src_total,
                    me_nn_stats_out *out);

/* One point-to-point ICP correspondence + reduction step: what an iteration of Open3D's
 * RegistrationICP(.., TransformationEstimationPointToPoint) (called at map_eval.cpp:1369-1371) needs from the clouds.
 * Over the correspondences of the last me_nn1(query_slot, ref) with d2 < max_distance^2 (Open3D SearchHybrid semantics):
 * the count, sum p, sum q, sum p q^T (row-major, p = source, q = target; all RELATIVE TO `origin`) and sum d2
 * (fitness = n_corr / n_source, inlier_rmse = sqrt(sum_d2 / n_corr)).  The 3x3 Umeyama / Kabsch solve stays on the host;
 * me_transform_cloud applies the update. */
typedef struct me_icp_sums {
    int64_t n_corr;
    int64_t n_source;
    double origin[3];
    double sum_p[3];
    double sum_q[3];
    double sum_pq[9];
    double sum_d2;
} me_icp_sums;
int me_icp_p2p_sums(me_ctx *ctx, int query_slot, double max_distance, me_icp_sums *out);

/* registration_methods 1 (point-to-plane) and 2 (generalized ICP, the shipped config's default; map_eval.cpp:1373-1384):
 * Open3D RegistrationICP(.., TransformationEstimationPointToPlane) / RegistrationGeneralizedICP [upstream].  Per-point
 * attributes live on the device in the caller's point order and follow the cloud through me_transform_cloud
 * (n <- R n, C <- R C R^T, as PointCloud::Transform) and me_voxel_downsample (normals averaged per voxel).
 *   me_set_normals       normals that came with the cloud (PCD normal_x/y/z), N x 3.
 *   me_get_normals       the current normals, N x 3.
 *   me_estimate_normals  PointCloud::EstimateNormals(KDTreeSearchParamKNN(knn)) of a cloud WITHOUT normals: exact k-NN of
 *                        every point (itself included), utility::ComputeCovariance, FastEigen3x3; (0,0,1) where
 *                        undefined.  1 <= knn <= 40.  Optional outputs: normals N x 3, and the neighbours themselves,
 *                        knn_idx / knn_d2 N x knn ascending by (d2, index), -1 / +inf where the cloud has fewer points.
 *   me_gicp_covariances  InitializePointCloudForGeneralizedICP(epsilon): C = Rx diag(epsilon,1,1) Rx^T from the normals
 *                        (estimated with knn = 20 when the slot has none); optional output N x 9 row-major.
 *   me_icp_lsq_sums      one correspondence + reduction step over the pairs of the last me_nn1(query_slot, ref) with
 *                        d2 < max_distance^2: J^T J (6x6 row-major), J^T r, sum r^2 of utility::ComputeJTJandJTr and
 *                        sum d2 (fitness = n_corr / n_source, inlier_rmse = sqrt(sum_d2 / n_corr)).  The host solves
 *                        JTJ x = -JTr, converts x with TransformVector6dToMatrix4d and calls me_transform_cloud.
 *                        ME_ICP_POINT_TO_PLANE needs normals on the ref cloud, ME_ICP_GENERALIZED covariances on both. */
#define ME_ICP_POINT_TO_PLANE 1
#define ME_ICP_GENERALIZED 2
typedef struct me_icp_lsq {
    int64_t n_corr;
    int64_t n_source;
    double JTJ[36];
    double JTr[6];
    double r2;
    double sum_d2;
} me_icp_lsq;
int me_set_normals(me_ctx *ctx, int slot, const double *normals);
int me_get_normals(me_ctx *ctx, int slot, double *normals);
int me_estimate_normals(me_ctx *ctx, int slot, int knn, double *normals, int32_t *knn_idx, double *knn_d2);
int me_gicp_covariances(me_ctx *ctx, int slot, double epsilon, double *cov);
int me_get_covariances(me_ctx *ctx, int slot, double *cov); /* the current N x 9 covariances (after any transform) */
int me_icp_lsq_sums(me_ctx *ctx, int query_slot, int mode, double max_distance, me_icp_lsq *out);

/* renderDistanceOnPointCloud (map_eval.cpp:586-607; raw_rendered_dis_map.pcd / inlier_rendered_dis_map.pcd, :485-495) for
 * the queries of the last me_nn1(query_slot, ...): rgb[N][3] in the caller's cloud order = ColorMapJet(min(d2, dis) / dis)
 * (the SQUARED distance against the unsquared `dis`, as the reference does; it repeats a serial KD-tree pass for it,
 * computePointCloudDistance :568-584 — the numbers are those of me_nn1).  inlier[N] (optional) = the gate of me_nn_stats,
 * i.e. the rows of corresponding_cloud_est (:1086-1087): their colours are the inlier rendering. */
int me_render_distance(me_ctx *ctx, int query_slot, double dis, double gate, int gate_mode, double *rgb, uint8_t *inlier);

/* ColorPointCloudByMME(pointcloud, entropies) (map_eval.cpp:686-735; map_entropy.pcd / gt_entropy.pcd) from the last
 * me_mme(slot): the VALID points in cloud order with their Jet colour of the log-mapped normalised |entropy|; range =
 * (|max|, |min|) over the non-zero entropies (:696-699).  xyz = rgb = NULL: count and range only. */
int me_render_entropy(me_ctx *ctx, int slot, double *xyz, double *rgb, int64_t capacity, int64_t *n_valid, double *min_abs,
                      double *max_abs);

/* computeChamferDistance (map_eval.cpp:1398-1431): both directions, no gate.  Runs me_nn1 both ways. */
int me_chamfer(me_ctx *ctx, double *cd);

/* ---- MME: ComputeMeanMapEntropyUsingNormalTBB / UsingNormal / ComputeMeanMapEntropy
 *      (map_eval.cpp:1608-1737, 1538-1606, 1438-1535) ------------------------------------------------------------ */
/* min_k: 10 for the estimated cloud (:1675), 5 for the GT cloud (:1458).  entropies[N] (0.0 where invalid) and
 * valid[N] are in cloud order, nullable.  sum_H / n_valid are shard-local partial sums; the mean entropy is
 * sum_H / n_valid (0 when n_valid == 0, :1720-1724). */
int me_mme(me_ctx *ctx, int slot, double radius, int min_k, double *entropies, uint8_t *valid, double *sum_H,
           int64_t *n_valid);
/* The per-point arrays of the slot's LAST MME pass (me_mme, me_run_suite, me_run_suite_from), as me_mme returns them: what
 * est_entropies / valid_entropy_points / gt_entropies hold after computeMME (map_eval.cpp:149-189).  Either may be NULL. */
int me_mme_fetch(me_ctx *ctx, int slot, double *entropies, uint8_t *valid);

/* ---- voxel Gaussians: VoxelCalculator::buildVoxelMap + computeVoxelEntropy (voxel_calculator.cpp:21-56,97-113) */
/* Output rows are in ascending (ix,iy,iz) order.  sigma is AS STORED by the reference after buildVoxelMap, i.e.
 * M2/(n-1)^2 for n > 10 and raw M2 otherwise (row-major 3x3).  *n_voxels in: capacity, out: count
 * (ME_ERR_CAPACITY if too small; pass all-NULL arrays to query the count). */
int me_voxel_gaussians(me_ctx *ctx, int slot, double voxel_size, int32_t *keys /*V x 3*/, int32_t *npts /*V*/,
                       double *mu /*V x 3*/, double *sigma /*V x 9*/, double *entropy /*V*/, int64_t *n_voxels);

/* ---- AWD + CDF + SCS: MapEval::calculateVMD (map_eval.cpp:240-390) with updateVoxelMap (voxel_calculator.cpp:
 *      142-172) and computeWassersteinDistanceGaussian (:115-140) ------------------------------------------------ */
/* rows: n x 27 doubles in the column order of voxel_errors.txt (map_eval.cpp:292-302), ascending key order;
 * w_sorted: ascending W (the CDF file's first column, :330-340); both nullable.  *n_rows in: capacity, out: count.
 * awd = mean W (NaN if no voxel qualifies, :324), scs as :347-389 (NaN if no voxel has a neighbour).
 * counts[3] = active / old / new voxel counts (voxel_calculator.cpp:170), nullable. */
int me_awd_scs(me_ctx *ctx, double voxel_size, int min_pts /*100, :280*/, int scs_radius /*5, :353*/,
               double *rows, double *w_sorted, int64_t *n_rows, double *awd, double *scs, int64_t counts[3]);

/* Batched VoxelCalculator::computeWassersteinDistanceGaussian(voxel1, voxel2) (voxel_calculator.hpp:69,
 * voxel_calculator.cpp:115-140) on caller-provided Gaussians: mu*[count][3], sigma*[count][9] (row-major, AS STORED
 * in VoxelInfo::sigma), n*[count] -> w[count].  Host pointers.  Lets the reference's own voxel_errors.txt be replayed
 * through the device kernel. */
int me_w2_batch(me_ctx *ctx, const double *mu1, const double *sigma1, const int32_t *n1, const double *mu2,
                const double *sigma2, const int32_t *n2, int64_t count, double *w);

/* SCS of a caller-provided sparse W table (map_eval.cpp:347-389): keys[n][3] voxel indices, w[n].  Host pointers. */
int me_scs_table(me_ctx *ctx, const int32_t *keys, const double *w, int64_t n, int scs_radius, double *scs);

/* ---- whole suite in one call (what MapEval::process runs between load and save, map_eval.cpp:52-85) -------- */
typedef struct me_suite_params {
    double icp_max_distance; /* Param::icp_max_distance_ */
    int gate_mode;           /* ME_GATE_* */
    double trunc[5];         /* Param::trunc_dist_ */
    double nn_radius;        /* Param::nn_radius_ */
    double vmd_voxel_size;   /* Param::vmd_voxel_size_ */
    int evaluate_mme;        /* Param::evaluate_mme_ */
    int evaluate_gt_mme;     /* Param::evaluate_gt_mme_ */
    int min_pts;             /* 100 */
    int scs_radius;          /* 5 */
} me_suite_params;

typedef struct me_suite_out {
    me_nn_stats_out est_gt; /* est_gt_results */
    me_nn_stats_out gt_est; /* gt_est_results (intended (gt_i, map_nn) pairing; see DESIGN.md deviation #4) */
    double full_chamfer;    /* full_chamfer_dist */
    double mme_est, mme_gt; /* mme_est / mme_gt */
    int64_t mme_est_valid, mme_gt_valid;
    double awd, scs;        /* vmd / scs_overall */
    int64_t n_w_voxels;
    double stage_ms[8];     /* host wall clock per stage (each ends with a stream sync): [1] nn est->gt, [2] nn gt->est,
                             * [3] statistics, [4] mme est, [5] mme gt, [6] voxel Gaussians + AWD + CDF + SCS, [7] the whole call;
                             * [0]: me_run_suite_from only (upload + index) */
} me_suite_out;

int me_run_suite(me_ctx *ctx, const me_suite_params *p, me_suite_out *out);

/* The same pass STARTING FROM THE TWO RAW CLOUDS — one call for everything MapEval::process() does between VoxelDownSample
 * (map_eval.cpp:38-39) and the result writers: upload + index of both clouds, computeMME(map_3d_, gt_3d_) (:56) on the map AS
 * LOADED, *map_3d_ = map_3d_->Transform(T) (:1206; T row-major 4x4, NULL or the identity = none), both 1-NN directions with the
 * AC / COM / CD statistics (:76), voxel Gaussians, AWD, CDF, SCS (:85).  The caller stays single-threaded (as process() is, :4).
 *   est / gt           double[n][3] — host memory, or device memory with ME_SUITE_DEVICE_INPUT; both NULL = run on the clouds
 *                      already uploaded to the two slots (e.g. after me_voxel_downsample).  With resident clouds T is applied to
 *                      the resident map IN PLACE, as :1206 overwrites map_3d_: a second call with the same T != identity
 *                      transforms it again (and computes the MME on the already transformed map) — upload afresh, or pass the
 *                      identity, to evaluate the same pose twice.  An index that is missing or does not fit nn_radius is rebuilt
 *                      on nn_radius' lattice before the stages (and the second lane) start: the call from resident clouds returns
 *                      what the call from the raw clouds returns, bit for bit.
 *   ME_SUITE_OVERLAP   two lanes: the calling thread drives `ctx`, an internal thread drives me_twin(ctx) on its own low-priority
 *                      stream — the ground truth is uploaded / indexed and both voxel tables are built UNDER the map's VALU-bound
 *                      MME kernel, and each lane searches one 1-NN direction (schedule: csrc/me_suite.hip).  Same kernels on the
 *                      same data: every result is bit-identical to the call without the flag and to me_run_suite.
 * Afterwards the per-point products are on the device as after the separate calls: me_mme_fetch, me_nn_fetch,
 * me_render_entropy / me_render_distance, me_awd_scs(rows, w_sorted) (cached tables), me_download_cloud (the transformed map).
 * stage_ms: wall clock of the calling thread per stage — [0] upload + index (+ transform) of the map (without ME_SUITE_OVERLAP: of
 * both clouds), [1] 1-NN map -> ground truth + partial sums, [2] the other direction (with ME_SUITE_OVERLAP: the wait for the second
 * lane), [3] sigma passes, [4] MME map, [5] MME ground truth, [6] AWD + CDF + SCS (+ voxel tables without the second lane),
 * [7] the whole call. */
#define ME_SUITE_OVERLAP 1
#define ME_SUITE_DEVICE_INPUT 2
#define ME_SUITE_PIN_HOST_INPUT 4 /* host input: page-lock the caller's two buffers for the duration of the call (hipHostRegister;
                                   * ~2 ms per 1.2 GB where measured), so that pageable memory — a std::vector, Open3D's points_ —
                                   * crosses PCIe at the pinned rate (21 instead of 30 ms per 50 M points); already pinned: no-op */
int me_run_suite_from(me_ctx *ctx, const double *est, int64_t n_est, const double *gt, int64_t n_gt, const double *T_rowmajor4x4,
                      const me_suite_params *p, int flags, me_suite_out *out);

/* ---- instrumentation (bench.py roofline leg) ------------------------------------------------------------- */
/* Average device time (ms, HIP events on the context's stream) and launch count of a named kernel family since
 * the last me_timers_reset: "nn_grid", "nn1", "mme", "sort", "morton", "gather", "cells" (the cell tables), "octree", "nn_stats", "voxel", "w2", "scs", "slab_filter", "halo_pack".  Enabled by me_timers_enable(1).
 * Counters (total_ms = 0, value in *launches): "mme_pairs" (accepted (query, neighbour) pairs of the MME launches: the useful work of
 * the VALU-bound kernel, bench.py's roofline.valu), "mme_refined" (queries whose thin neighbourhood — smallest covariance eigenvalue below ~1.8e-6 cell^2 — the MME pass
 * recomputed two-pass about the query itself; counted whether or not timers are on), "nn_queries" / "nn_fallback_queries" (1-NN queries, and those that needed the
 * octree pass), "nn1_opened" / "nn1_scans" / "nn1_points" / "nn1_max_opened" (octree walk: nodes opened, leaf cells and points
 * scanned, the longest chain of one query in the octet walk), "nn1_far" (walks handed over to the wave-per-query kernel). */
int me_timers_enable(me_ctx *ctx, int on);
int me_timers_reset(me_ctx *ctx);
int me_timer_get(me_ctx *ctx, const char *name, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
