// map_eval_dist.cpp — `num_gpus: N` for the drop-in binary: MapEval::process() between "clouds loaded and down-sampled"
// and "results written" (map_eval/src/map_eval.cpp:51-85) sharded over N GPUs, one process per GPU, collectives over RCCL
// (dist_comm.hpp).  C++ restatement of cloud_map_evaluation_amd/dist.py::suite_step_slab on the library's slab entry points
// (include/mapeval_hip.h: me_set_slab, me_nn_unresolved / me_nn_points_covered / me_nn_patch, me_nn_partial_sums /
// me_nn_sigma_sums / me_nn_finalize, me_voxel_partial_rows_device / me_voxel_merge_device, me_slab_points):
//
//   every rank starts from ITS 1/N of each cloud's points (the r-th contiguous piece: what it would have read of the files; this
//   host reads whole files and uses its piece) and the ranks exchange them once:
//     cuts     equal-count slab faces along the longest axis of the ground truth from ONE all-gather of a fixed-size sample
//     halo     me_halo_pack_tagged_device copies every local point (+ its index in the whole cloud) into the send segment of every
//              rank whose slab + halo holds it; the segment sizes go round in one all-gather, the points and tags in grouped
//              ncclSend / ncclRecv (Comm::all_to_all_v); what arrives IS the rank's slab + halo (me_upload_slab_device)
//     MME      per-point on the slab, {sum H, n_valid} all-reduced                        (1 collective, 4 doubles)
//     AC/COM   local 1-NN both ways; queries that a closer point on another rank could beat go through the cross-rank
//              step: counts all-reduce, queries + bounds all-gather, bounded search on every rank (outside the band the owner has
//              searched: me_nn_points_covered), MIN all-reduce, patch;
//              partial sums all-reduce, sigma numerators all-reduce
//     AWD/SCS  voxel partial rows of the owned points: counts ride on the sums, one padded all-gather per cloud, Chan
//              merge on the device, me_awd_scs on the merged (replicated) tables
//     outputs  (only when files are written) the owned entries of per-point entropies / squared distances travel to rank 0 as
//              (tag, value) pairs — every rank sends 1/N of the cloud, nobody moves whole-cloud arrays — and rank 0 hands them to
//              a second, whole-cloud context (me_set_mme_result / me_set_nn_result) that writes every file as the single-GPU path does.
//   Registration (evaluate_using_initial: false — what every shipped config asks for, config.yaml:2,53): the loop of
//   performRegistration (map_eval.cpp:191-237, :1366-1394) runs on the replicated clouds with the QUERIES sharded (me_set_shard):
//   per iteration a local me_nn1 on 1/N of the map, the additive sums of me_icp_p2p_sums / me_icp_lsq_sums all-reduced (17 / 30
//   doubles), the same 4x4 / 6x6 solve on every rank, me_transform_cloud; then the slab pipeline above on the registered map with
//   the ICP path's gate (d2 < max^2, :1168).  With a non-identity initial_matrix (or after registration) the map is transformed
//   BEFORE the MME pass (the reference transforms it after, :1206): for a rigid matrix the entropies agree to rounding.
//   A rank that fails says so at the next agreement point (Comm::all_ok) or exits: the launcher (map_eval_main.cpp) takes the
//   other ranks down with it instead of leaving them in a collective.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <limits>
#include <utility>
#include <vector>

#include "dist_comm.hpp"
#include "map_eval.h"

namespace {

constexpr int kPart = 2 + 5 + 5 + 5 + 1;  // me_nn_partial as doubles

void pack(const me_nn_partial &p, double *v) {
    v[0] = (double) p.n_query;
    v[1] = (double) p.n_corr;
    for (int k = 0; k < 5; ++k) {
        v[2 + k] = (double) p.n_inl[k];
        v[7 + k] = p.sum_d[k];
        v[12 + k] = p.sum_d2[k];
    }
    v[17] = p.sum_sqrt_all;
}
void unpack(const double *v, me_nn_partial &p) {
    p.n_query = (int64_t) std::llround(v[0]);
    p.n_corr = (int64_t) std::llround(v[1]);
    for (int k = 0; k < 5; ++k) {
        p.n_inl[k] = (int64_t) std::llround(v[2 + k]);
        p.sum_d[k] = v[7 + k];
        p.sum_d2[k] = v[12 + k];
    }
    p.sum_sqrt_all = v[17];
}

struct TicToc3 {  // milliseconds since construction (include/tic_toc.h:10-24)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double toc() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

bool h2d(void *d, const void *h, size_t n) { return n == 0 || hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; }
bool d2h(void *h, const void *d, size_t n) { return n == 0 || hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess; }

}  // namespace

#define DIST_TRY(expr)                                     \
    do {                                                   \
        if ((expr) != ME_OK) return fail(me_last_error(ctx_)); \
    } while (0)
#define COMM_TRY(expr)                                                          \
    do {                                                                        \
        if (!(expr)) return fail(std::string("collective failed: ") + comm_->err); \
    } while (0)

// all-reduce of a small host vector of doubles
int MapEval::allReduceHost(std::vector<double> &v, bool min_op) {
    medist::DevMem &buf = pool_[6];  // (pooled: one hipMalloc for the run, not one per collective)
    if (!buf.ensure(v.size() * 8) || !h2d(buf.p, v.data(), v.size() * 8)) return fail("device staging buffer");
    COMM_TRY(min_op ? comm_->all_reduce_min_f64(buf.as<double>(), v.size()) : comm_->all_reduce_sum_f64(buf.as<double>(), v.size()));
    if (!d2h(v.data(), buf.p, v.size() * 8)) return fail("device staging buffer");
    return 0;
}

// the registration step's sums are additive over the shards of the map (me_icp_p2p_sums / me_icp_lsq_sums under me_set_shard):
// one all-reduce of 17 / 30 doubles per iteration, after which every rank holds the whole-cloud sums and solves the same system
int MapEval::reduceIcp(me_icp_sums &s, me_icp_lsq &q, int method) {
    std::vector<double> v;
    if (method == 0) {
        v.assign(17, 0.0);
        v[0] = (double) s.n_corr;
        for (int k = 0; k < 3; ++k) {
            v[1 + k] = s.sum_p[k];
            v[4 + k] = s.sum_q[k];
        }
        for (int k = 0; k < 9; ++k) v[7 + k] = s.sum_pq[k];
        v[16] = s.sum_d2;
        if (allReduceHost(v, false) != 0) return -1;
        s.n_corr = (int64_t) std::llround(v[0]);
        for (int k = 0; k < 3; ++k) {
            s.sum_p[k] = v[1 + k];
            s.sum_q[k] = v[4 + k];
        }
        for (int k = 0; k < 9; ++k) s.sum_pq[k] = v[7 + k];
        s.sum_d2 = v[16];
    } else {
        v.assign(1 + 36 + 6 + 2, 0.0);
        v[0] = (double) q.n_corr;
        for (int k = 0; k < 36; ++k) v[1 + k] = q.JTJ[k];
        for (int k = 0; k < 6; ++k) v[37 + k] = q.JTr[k];
        v[43] = q.r2;
        v[44] = q.sum_d2;
        if (allReduceHost(v, false) != 0) return -1;
        q.n_corr = (int64_t) std::llround(v[0]);
        for (int k = 0; k < 36; ++k) q.JTJ[k] = v[1 + k];
        for (int k = 0; k < 6; ++k) q.JTr[k] = v[37 + k];
        q.r2 = v[43];
        q.sum_d2 = v[44];
        s.n_corr = q.n_corr;
        s.sum_d2 = q.sum_d2;
    }
    return 0;
}

// owned entries of a per-point array of this rank's slab -> rank 0's whole-cloud array, as (tag, value[, flag]) records: a rank
// sends what it owns (1/N of the cloud), nothing else moves.  tags[i] = index in the whole cloud of the i-th point this rank
// uploaded (it came with the point through the exchange); the per-point outputs are in upload order (me_slab_points: identity).
int MapEval::gatherPerPoint(int slot, size_t n_global, const std::vector<int64_t> &tags, const std::vector<double> *vals,
                            const std::vector<uint8_t> *flags, std::vector<double> *vals_out, std::vector<uint8_t> *flags_out) {
    const int world = comm_->world, rank = comm_->rank;
    int64_t n_loc = 0;
    DIST_TRY(me_slab_points(ctx_, slot, nullptr, nullptr, 0, &n_loc));
    std::vector<uint8_t> owned((size_t) n_loc);
    if (n_loc) DIST_TRY(me_slab_points(ctx_, slot, nullptr, owned.data(), n_loc, &n_loc));
    // records of three doubles: tag, value, flag
    std::vector<double> rec;
    rec.reserve((size_t) n_loc * 3 / (size_t) std::max(1, world) + 64);
    for (int64_t i = 0; i < n_loc; ++i)
        if (owned[(size_t) i]) {
            rec.push_back((double) tags[(size_t) i]);
            rec.push_back(vals ? (*vals)[(size_t) i] : 0.0);
            rec.push_back(flags ? (double) (*flags)[(size_t) i] : 0.0);
        }
    // sizes: every rank tells every rank how much it sends to rank 0
    std::vector<double> cnt((size_t) world, 0.0);
    cnt[(size_t) rank] = (double) rec.size();
    if (allReduceHost(cnt, false) != 0) return -1;
    std::vector<size_t> sb((size_t) world, 0), rb((size_t) world, 0);
    sb[0] = rec.size() * 8;
    size_t total = 0;
    if (rank == 0)
        for (int k = 0; k < world; ++k) {
            rb[(size_t) k] = (size_t) cnt[(size_t) k] * 8;
            total += rb[(size_t) k];
        }
    medist::DevMem &snd = pool_[4], &rcv = pool_[5];
    if (!snd.ensure(rec.size() * 8 + 8) || !rcv.ensure(total + 8) || !h2d(snd.p, rec.data(), rec.size() * 8)) return fail("device staging buffer");
    COMM_TRY(comm_->all_to_all_v(snd.p, sb.data(), rcv.p, rb.data()));
    if (rank != 0) return 0;
    std::vector<double> all(total / 8);
    if (!d2h(all.data(), rcv.p, total)) return fail("device staging buffer");
    if (vals_out) vals_out->assign(n_global, 0.0);
    if (flags_out) flags_out->assign(n_global, 0);
    for (size_t r = 0; r + 2 < all.size(); r += 3) {
        const size_t g = (size_t) all[r];
        if (g >= n_global) return fail("per-point gather: tag out of range");
        if (vals_out) (*vals_out)[g] = all[r + 1];
        if (flags_out) (*flags_out)[g] = (uint8_t) all[r + 2];
    }
    return 0;
}

// The one-shot halo exchange of one cloud: this rank's piece [i0, i1) of the whole cloud `pts` (host) -> its slab + halo on the
// device (recv: xyz, tags_out: index in the whole cloud of every received point).
int MapEval::exchangeCloud(const std::vector<double> &pts, size_t i0, size_t i1, const double *T, int axis, const std::vector<double> &cuts,
                           double halo, medist::DevMem &recv, std::vector<int64_t> &tags_out, int64_t *n_recv) {
    const int world = comm_->world, rank = comm_->rank;
    const int64_t n_loc = (int64_t) (i1 - i0);
    medist::DevMem &piece = pool_[0], &packed = pool_[1], &tagbuf = pool_[2], &tagrecv = pool_[3];
    if (!piece.ensure((size_t) n_loc * 24 + 8) || !h2d(piece.p, pts.data() + 3 * i0, (size_t) n_loc * 24)) return fail("device buffer of the local piece");
    // *map_3d_ = map_3d_->Transform(T) (:1206) on the piece, before slab membership is decided
    if (T && n_loc) DIST_TRY(me_transform_points_device(ctx_, piece.as<double>(), n_loc, T));
    std::vector<int64_t> counts((size_t) world, 0);
    DIST_TRY(me_halo_pack_tagged_device(ctx_, piece.as<double>(), n_loc, axis, cuts.data(), world, halo, nullptr, nullptr, 0, 0, counts.data()));
    int64_t total = 0;
    for (int k = 0; k < world; ++k) total += counts[(size_t) k];
    if (!packed.ensure((size_t) total * 24 + 8) || !tagbuf.ensure((size_t) total * 8 + 8)) return fail("device buffers of the halo exchange");
    if (total)
        DIST_TRY(me_halo_pack_tagged_device(ctx_, piece.as<double>(), n_loc, axis, cuts.data(), world, halo, packed.as<double>(),
                                            tagbuf.as<int64_t>(), (int64_t) i0, total, counts.data()));
    // the segment sizes go round (world x world numbers), then the points and their tags
    std::vector<double> table((size_t) world * (size_t) world, 0.0);
    for (int k = 0; k < world; ++k) table[(size_t) rank * (size_t) world + (size_t) k] = (double) counts[(size_t) k];
    if (allReduceHost(table, false) != 0) return -1;
    std::vector<size_t> sb((size_t) world), rb((size_t) world), sb8((size_t) world), rb8((size_t) world);
    int64_t got = 0;
    for (int k = 0; k < world; ++k) {
        const size_t c_out = (size_t) counts[(size_t) k], c_in = (size_t) table[(size_t) k * (size_t) world + (size_t) rank];
        sb[(size_t) k] = c_out * 24;
        rb[(size_t) k] = c_in * 24;
        sb8[(size_t) k] = c_out * 8;
        rb8[(size_t) k] = c_in * 8;
        got += (int64_t) c_in;
    }
    if (!recv.ensure((size_t) got * 24 + 8) || !tagrecv.ensure((size_t) got * 8 + 8)) return fail("device buffers of the halo exchange");
    COMM_TRY(comm_->all_to_all_v(packed.p, sb.data(), recv.p, rb.data()));
    COMM_TRY(comm_->all_to_all_v(tagbuf.p, sb8.data(), tagrecv.p, rb8.data()));
    tags_out.resize((size_t) got);
    if (!d2h(tags_out.data(), tagrecv.p, (size_t) got * 8)) return fail("device buffers of the halo exchange");
    *n_recv = got;
    return 0;
}

int MapEval::processDist(double t_loaded) {
    const int rank = comm_->rank, world = comm_->world;
    const size_t n_e = map_3d_->size(), n_g = gt_3d_->size();
    t1 = t_loaded;
    // wall-clock split of the run (rank 0 prints it at the end): a phase ends when its last call has returned — the C ABI's
    // calls return their scalars, so nothing of a phase is still in flight at its mark
    TicToc3 phase_clock;
    double phase_last = 0.0;
    std::vector<std::pair<const char *, double>> phases;
    auto mark = [&](const char *name) {
        const double now = phase_clock.toc();
        phases.emplace_back(name, now - phase_last);
        phase_last = now;
    };
    // ---- the map in its final pose: initial_matrix (:1206), or the registration result (performRegistration, :191-237) ----
    int gate_mode = ME_GATE_LE_UNSQUARED;  // (:1219, sic) — the ICP path gates d2 < max^2 (:1168)
    if (!param_.evaluate_using_initial_) {
        if (param_.evaluation_method_ < 0 || param_.evaluation_method_ > 2) return fail("Invalid registration type specified");  // (:1385-1387)
        // the whole clouds are resident on every rank (process() uploaded them for VoxelDownSample): queries sharded, sums all-reduced
        if (performRegistration(/*metrics=*/false) != 0) return -1;  // leaves the registered map in map_3d_ on every rank
        gate_mode = ME_GATE_LT_SQUARED;
    } else {
        bool identity = true;
        for (int i = 0; i < 16; ++i) identity = identity && (param_.initial_matrix_[i] == ((i % 5 == 0) ? 1.0 : 0.0));
        if (!identity) {
            DIST_TRY(me_transform_cloud(ctx_, ME_SLOT_EST, param_.initial_matrix_.data()));
            DIST_TRY(me_download_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data()));
        }
    }
    // (test hook: MAPEVAL_TEST_FAIL_RANK=r makes rank r fail here, tests/test_gpu_host.py)
    if (const char *fr = std::getenv("MAPEVAL_TEST_FAIL_RANK"))
        if (std::atoi(fr) == rank) fail("MAPEVAL_TEST_FAIL_RANK: this rank was told to fail");
    if (!comm_->all_ok(last_error.empty())) return fail("a rank failed before the exchange: stopping");
    mark(param_.evaluate_using_initial_ ? "pose" : "registration");
    // ---- this rank's piece of each cloud (distributed input) ----
    auto piece = [&](size_t n, size_t &i0, size_t &i1) {
        i0 = (size_t) ((unsigned long long) n * (unsigned long long) rank / (unsigned long long) world);
        i1 = (size_t) ((unsigned long long) n * (unsigned long long) (rank + 1) / (unsigned long long) world);
    };
    size_t e0, e1, g0, g1;
    piece(n_e, e0, e1);
    piece(n_g, g0, g1);
    // ---- slab faces: equal-count cuts along the longest axis of the ground truth, from ONE all-gather of a fixed-size strided
    // sample of every rank's piece (identical on every rank; the same recipe as dist.py::slab_cuts_from_sample) ----
    constexpr size_t kSample = 16384;
    int axis = 0;
    std::vector<double> cuts((size_t) world + 1);
    {
        std::vector<double> mine(kSample * 3, std::nan(""));
        const size_t n_loc = g1 - g0, take = std::min(kSample, n_loc);
        for (size_t j = 0; j < take; ++j) {
            const size_t i = g0 + (size_t) ((unsigned long long) j * (unsigned long long) n_loc / (unsigned long long) take);
            for (int d = 0; d < 3; ++d) mine[3 * j + (size_t) d] = gt_3d_->points_[3 * i + (size_t) d];
        }
        medist::DevMem &sm = pool_[4], &sa = pool_[5];
        if (!sm.ensure(kSample * 24) || !sa.ensure(kSample * 24 * (size_t) world) || !h2d(sm.p, mine.data(), kSample * 24)) return fail("device buffers of the sample");
        COMM_TRY(comm_->all_gather(sm.p, sa.p, kSample * 24));
        std::vector<double> all(kSample * 3 * (size_t) world);
        if (!d2h(all.data(), sa.p, all.size() * 8)) return fail("device buffers of the sample");
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        std::vector<double> coord[3];
        for (size_t i = 0; i < all.size() / 3; ++i) {
            if (std::isnan(all[3 * i])) continue;
            for (int d = 0; d < 3; ++d) {
                lo[d] = std::min(lo[d], all[3 * i + (size_t) d]);
                hi[d] = std::max(hi[d], all[3 * i + (size_t) d]);
                coord[d].push_back(all[3 * i + (size_t) d]);
            }
        }
        for (int d = 1; d < 3; ++d)
            if (hi[d] - lo[d] > hi[axis] - lo[axis]) axis = d;
        std::vector<double> &c = coord[axis];
        std::sort(c.begin(), c.end());
        cuts[0] = -INFINITY;
        cuts[(size_t) world] = INFINITY;
        for (int k = 1; k < world; ++k) {
            cuts[(size_t) k] = c.empty() ? (double) k : c[std::min(c.size() - 1, c.size() * (size_t) k / (size_t) world)];
            if (!(cuts[(size_t) k] > cuts[(size_t) k - 1])) cuts[(size_t) k] = std::nextafter(cuts[(size_t) k - 1], INFINITY);
        }
    }
    mark("cuts");
    const double halo = std::max(1.0, 1.0001 * param_.nn_radius_);  // MME needs halo >= nn_radius; 1 m covers the usual 1-NN reach
    // ---- one-shot halo exchange, then the slab upload (what arrived IS slab + halo: no filter pass) ----
    medist::DevMem recv_e, recv_g;
    std::vector<int64_t> tags_e, tags_g;
    int64_t got_e = 0, got_g = 0;
    if (exchangeCloud(map_3d_->points_, e0, e1, nullptr, axis, cuts, halo, recv_e, tags_e, &got_e) != 0) return -1;
    if (exchangeCloud(gt_3d_->points_, g0, g1, nullptr, axis, cuts, halo, recv_g, tags_g, &got_g) != 0) return -1;
    mark("halo_exchange");
    DIST_TRY(me_set_slab(ctx_, axis, cuts[(size_t) rank], cuts[(size_t) rank + 1], halo));
    // (round 6: the index builds also emit the voxel run records, so that the voxel partial rows below have no pass over the slab left)
    DIST_TRY(me_set_voxel_hint(ctx_, param_.vmd_voxel_size_));
    DIST_TRY(me_upload_slab_device(ctx_, ME_SLOT_EST, recv_e.as<double>(), got_e, param_.nn_radius_));
    DIST_TRY(me_upload_slab_device(ctx_, ME_SLOT_GT, recv_g.as<double>(), got_g, param_.nn_radius_));
    DIST_TRY(me_set_voxel_hint(ctx_, 0.0));
    mark("index");
    if (rank == 0)
        std::cout << "INFO: multi-GPU run: " << world << " rank(s) over " << comm_->name() << ", slabs along axis " << axis << ", halo "
                  << halo << " m; rank 0 holds " << got_e << " + " << got_g << " of " << n_e << " + " << n_g << " points" << std::endl;
    // rank 0 keeps a second context with the WHOLE clouds for the colour renderers (only when files are written)
    if (rank == 0 && param_.save_immediate_result_) {
        render_ctx_ = me_create(param_.gpu_device, 0);
        if (!render_ctx_) return fail("GPU engine unavailable for the renderers");
        if (me_upload_cloud(render_ctx_, ME_SLOT_EST, map_3d_->points_.data(), (int64_t) n_e, nullptr, param_.nn_radius_) != ME_OK ||
            me_upload_cloud(render_ctx_, ME_SLOT_GT, gt_3d_->points_.data(), (int64_t) n_g, nullptr, param_.nn_radius_) != ME_OK)
            return fail(me_last_error(render_ctx_));
    }
    mark("render_context");
    TicToc3 clock;

    // ---- MME (computeMME, :149-189): est k >= 10, gt k >= 5 ----
    std::vector<double> sums(4, 0.0);
    if (param_.evaluate_mme_) {
        const bool want_points = param_.save_immediate_result_;  // per-point arrays only feed the renderers / entropy files
        for (int pass = 0; pass < (param_.evaluate_gt_mme_ ? 2 : 1); ++pass) {
            const int slot = pass == 0 ? ME_SLOT_EST : ME_SLOT_GT;
            const int64_t n_loc = me_cloud_size(ctx_, slot);
            std::vector<double> ent(want_points ? (size_t) n_loc : 0, 0.0);
            std::vector<uint8_t> val(want_points ? (size_t) n_loc : 0, 0);
            double s = 0;
            int64_t nv = 0;
            DIST_TRY(me_mme(ctx_, slot, param_.nn_radius_, pass == 0 ? 10 : 5, want_points ? ent.data() : nullptr,
                            want_points ? val.data() : nullptr, &s, &nv));
            sums[(size_t) 2 * pass] = s;
            sums[(size_t) 2 * pass + 1] = (double) nv;
            if (!want_points) continue;
            std::vector<uint8_t> gv;
            if (gatherPerPoint(slot, pass == 0 ? n_e : n_g, pass == 0 ? tags_e : tags_g, &ent, &val, pass == 0 ? &est_entropies : &gt_entropies,
                               pass == 0 ? &valid_entropy_points : &gv) != 0)
                return -1;
            if (render_ctx_) {
                if (me_set_mme_result(render_ctx_, slot, (pass == 0 ? est_entropies : gt_entropies).data(),
                                      (pass == 0 ? valid_entropy_points : gv).data()) != ME_OK)
                    return fail(me_last_error(render_ctx_));
                std::swap(ctx_, render_ctx_);  // renderEntropy works on ctx_
                const bool ok = renderEntropy(slot, pass == 0 ? map_entropy_xyz : gt_entropy_xyz, pass == 0 ? map_entropy_rgb : gt_entropy_rgb, true);
                std::swap(ctx_, render_ctx_);
                if (!ok) return -1;
            }
        }
        if (allReduceHost(sums, false) != 0) return -1;
        mme_est = sums[1] > 0 ? sums[0] / sums[1] : 0.0;
        mme_gt = sums[3] > 0 ? sums[2] / sums[3] : 0.0;
        if (rank == 0) {
            if (param_.evaluate_gt_mme_) std::cout << "MME EST-GT: " << mme_est << " " << mme_gt << std::endl;
            else std::cout << "MME EST: " << mme_est << std::endl;
            if (param_.save_immediate_result_) saveMmeResults();
        }
    }
    t2 = t1 + clock.toc();
    mark("mme");

    // ---- AC / COM / CD (calculateMetricsWithInitialMatrix, :1204-1260) ----
    const int dirs[2][2] = {{ME_SLOT_EST, ME_SLOT_GT}, {ME_SLOT_GT, ME_SLOT_EST}};
    int64_t cnt[2] = {0, 0};
    for (int d = 0; d < 2; ++d) {
        DIST_TRY(me_nn1(ctx_, dirs[d][0], dirs[d][1], nullptr, nullptr));
        DIST_TRY(me_nn_unresolved(ctx_, dirs[d][0], nullptr, nullptr, 0, &cnt[d]));
    }
    {   // cross-rank step: open queries of both directions in one all-gather, their answers in one MIN-reduce
        // Round 5: ONE fixed-capacity message per rank, written / answered / patched by three library calls (me_nn_cross_message,
        // me_nn_cross_answer, me_nn_cross_patch); the open-query counts of all ranks ride on the messages' header rows (no separate
        // all-reduce), the bands every owner has searched are expanded on the device from the cuts.  A rank with more open queries
        // than the capacity makes everybody take the exactly sized form below (round 3).
        constexpr int64_t kCrossCap = 4096;
        const size_t rows = 1 + 2 * (size_t) kCrossCap;
        std::vector<double> table((size_t) world * 2, 0.0);
        bool lean_done = false;
        if (world > 1) {
            medist::DevMem &msg = pool_[0], &all = pool_[1], &ans = pool_[2];  // (the exchange is over: its buffers are free)
            if (!msg.ensure(rows * 32) || !all.ensure(rows * 32 * (size_t) world) || !ans.ensure(rows * 8 * (size_t) world))
                return fail("device buffers of the cross-rank step");
            int64_t c2[2] = {0, 0};
            DIST_TRY(me_nn_cross_message(ctx_, msg.as<double>(), kCrossCap, 0, 0, c2));
            COMM_TRY(comm_->all_gather(msg.p, all.p, rows * 32));
            for (int k = 0; k < world; ++k) {
                double head[4];
                if (hipMemcpy(head, all.as<double>() + (size_t) k * rows * 4, sizeof head, hipMemcpyDeviceToHost) != hipSuccess) return fail("hipMemcpy");
                table[(size_t) k * 2] = head[0];
                table[(size_t) k * 2 + 1] = head[1];
            }
            int64_t cmax_all = 0, others[2] = {0, 0};
            for (int k = 0; k < world; ++k)
                for (int d = 0; d < 2; ++d) {
                    cmax_all = std::max<int64_t>(cmax_all, (int64_t) table[(size_t) k * 2 + d]);
                    if (k != rank) others[d] += (int64_t) table[(size_t) k * 2 + d];
                }
            if (cmax_all <= kCrossCap) {
                if (cmax_all > 0) {
                    const int mask = (others[0] > 0 ? 1 : 0) | (others[1] > 0 ? 2 : 0);
                    DIST_TRY(me_nn_cross_answer(ctx_, all.as<double>(), world, kCrossCap, rank, mask, axis, cuts.data(), halo, ans.as<double>()));
                    COMM_TRY(comm_->all_reduce_min_f64(ans.as<double>(), rows * (size_t) world));
                    DIST_TRY(me_nn_cross_patch(ctx_, ans.as<double>(), kCrossCap, rank));
                }
                lean_done = true;
            }
        }
        int64_t cmax[2] = {0, 0}, total = 0;
        for (int k = 0; k < world; ++k)
            for (int d = 0; d < 2; ++d) {
                cmax[d] = std::max<int64_t>(cmax[d], (int64_t) table[(size_t) k * 2 + d]);
                total += (int64_t) table[(size_t) k * 2 + d];
            }
        if (total > 0 && world > 1 && !lean_done) {
            // message of a rank: [X0 (cmax0 x 3) | D0 (cmax0) | X1 (cmax1 x 3) | D1 (cmax1)] doubles
            const size_t msg_len = (size_t) (cmax[0] + cmax[1]) * 4;
            const size_t off_x[2] = {0, (size_t) cmax[0] * 4}, off_d[2] = {(size_t) cmax[0] * 3, (size_t) cmax[0] * 4 + (size_t) cmax[1] * 3};
            medist::DevMem &msg = pool_[0], &all = pool_[1], &ans = pool_[2];  // (the exchange is over: its buffers are free)
            if (!msg.ensure(msg_len * 8) || !all.ensure(msg_len * 8 * (size_t) world) || !ans.ensure((size_t) (cmax[0] + cmax[1]) * 8 * (size_t) world))
                return fail("device buffers of the cross-rank step");
            // (hipMemset / device-to-device hipMemcpy on the null stream may return before they are done, and the library works on
            //  its own non-blocking stream: settle the NULL STREAM — not the device — before handing the buffer over)
            if (hipMemset(msg.p, 0, msg_len * 8) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return fail("hipMemset");
            for (int d = 0; d < 2; ++d)
                if (cnt[d]) DIST_TRY(me_nn_unresolved(ctx_, dirs[d][0], msg.as<double>() + off_x[d], msg.as<double>() + off_d[d], cnt[d], &cnt[d]));
            COMM_TRY(comm_->all_gather(msg.p, all.p, msg_len * 8));
            const size_t row = (size_t) (cmax[0] + cmax[1]);
            {
                std::vector<double> inf(row * (size_t) world, std::numeric_limits<double>::infinity());
                if (!h2d(ans.p, inf.data(), inf.size() * 8)) return fail("device buffers of the cross-rank step");
            }
            for (int k = 0; k < world; ++k)
                for (int d = 0; d < 2; ++d) {
                    const int64_t c = (int64_t) table[(size_t) k * 2 + d];
                    if (c == 0) continue;
                    double *dst = ans.as<double>() + (size_t) k * row + (d == 0 ? 0 : (size_t) cmax[0]);
                    const double *src = all.as<double>() + (size_t) k * msg_len;
                    // the bound to beat = the owner's own result; every other rank may lower it
                    if (hipMemcpy(dst, src + off_d[d], (size_t) c * 8, hipMemcpyDeviceToDevice) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
                        return fail("hipMemcpy");
                    if (k != rank) {
                        // rank k has searched its slab +- halo completely (it holds every point of both clouds in that band):
                        // only what sticks out of the band can beat its bound (me_nn_points_covered)
                        medist::DevMem &cov = pool_[3];  // (the tag buffer of the exchange: free by now)
                        std::vector<double> band((size_t) c * 2);
                        for (int64_t j = 0; j < c; ++j) {
                            band[(size_t) 2 * j] = cuts[(size_t) k] - halo;
                            band[(size_t) 2 * j + 1] = cuts[(size_t) k + 1] + halo;
                        }
                        if (!cov.ensure(band.size() * 8) || !h2d(cov.p, band.data(), band.size() * 8)) return fail("device buffers of the cross-rank step");
                        DIST_TRY(me_nn_points_covered(ctx_, dirs[d][1], src + off_x[d], c, dst, axis, cov.as<double>()));
                    }
                }
            COMM_TRY(comm_->all_reduce_min_f64(ans.as<double>(), row * (size_t) world));
            for (int d = 0; d < 2; ++d)
                if (cnt[d]) DIST_TRY(me_nn_patch(ctx_, dirs[d][0], ans.as<double>() + (size_t) rank * row + (d == 0 ? 0 : (size_t) cmax[0]), cnt[d]));
        }
    }
    // partial sums of both directions + the voxel row counts in one all-reduce; then the sigma numerators
    int64_t vrows[2] = {0, 0};
    for (int s = 0; s < 2; ++s)
        DIST_TRY(me_voxel_partial_rows_device(ctx_, s == 0 ? ME_SLOT_EST : ME_SLOT_GT, param_.vmd_voxel_size_, nullptr, 0, &vrows[s]));
    std::vector<double> vec((size_t) 2 * kPart + (size_t) 2 * world, 0.0);
    for (int d = 0; d < 2; ++d) {
        me_nn_partial p;
        DIST_TRY(me_nn_partial_sums(ctx_, dirs[d][0], param_.icp_max_distance_, gate_mode, param_.trunc_dist_.data(), &p));
        pack(p, vec.data() + (size_t) d * kPart);
        if (param_.enable_debug)
            std::cerr << "[rank " << rank << "] direction " << d << ": " << p.n_query << " owned queries, " << p.n_corr << " gated, "
                      << cnt[d] << " through the cross-rank step, sum sqrt(d2) = " << p.sum_sqrt_all << std::endl;
    }
    vec[(size_t) 2 * kPart + (size_t) rank] = (double) vrows[0];
    vec[(size_t) 2 * kPart + (size_t) world + (size_t) rank] = (double) vrows[1];
    if (allReduceHost(vec, false) != 0) return -1;
    me_nn_partial tot[2];
    std::vector<double> sig(10, 0.0);
    for (int d = 0; d < 2; ++d) {
        unpack(vec.data() + (size_t) d * kPart, tot[d]);
        double mean[5];
        for (int k = 0; k < 5; ++k) mean[k] = tot[d].n_corr > 0 ? tot[d].sum_d[k] / (double) tot[d].n_corr : 0.0;
        DIST_TRY(me_nn_sigma_sums(ctx_, dirs[d][0], param_.icp_max_distance_, gate_mode, mean, sig.data() + 5 * d));
    }
    if (allReduceHost(sig, false) != 0) return -1;
    me_nn_stats_out eg, ge;
    me_nn_finalize(&tot[0], sig.data(), (int64_t) n_e, &eg);
    me_nn_finalize(&tot[1], sig.data() + 5, (int64_t) n_g, &ge);
    if (param_.evaluate_using_initial_) finishInitialMatrixMetrics(eg, ge, clock.toc() / 1000.0);
    else finishRegistrationMetrics(eg, ge, clock.toc() / 1000.0);  // calculateMetrics' tail (:1147-1202)
    // squared distances of the map's points for raw_rendered_dis_map.pcd / inlier_rendered_dis_map.pcd (:485-495)
    if (param_.save_immediate_result_) {
        const int64_t n_loc = me_cloud_size(ctx_, ME_SLOT_EST);
        std::vector<double> d2((size_t) n_loc), d2_all;
        if (n_loc) DIST_TRY(me_nn_fetch(ctx_, ME_SLOT_EST, nullptr, d2.data()));
        if (gatherPerPoint(ME_SLOT_EST, n_e, tags_e, &d2, nullptr, &d2_all, nullptr) != 0) return -1;
        if (render_ctx_ && me_set_nn_result(render_ctx_, ME_SLOT_EST, ME_SLOT_GT, d2_all.data()) != ME_OK) return fail(me_last_error(render_ctx_));
    }
    t5 = t4 = t3 = t1 + clock.toc();
    mark("nn_metrics");

    // ---- AWD / CDF / SCS (calculateVMD, :240-390): Chan merge of every rank's voxel partials, then the replicated tables ----
    for (int s = 0; s < 2; ++s) {
        const int slot = s == 0 ? ME_SLOT_EST : ME_SLOT_GT;
        int64_t vmax = 1;
        for (int k = 0; k < world; ++k) vmax = std::max<int64_t>(vmax, (int64_t) std::llround(vec[(size_t) 2 * kPart + (size_t) s * world + (size_t) k]));
        medist::DevMem &mine = pool_[0], &all = pool_[1];
        if (!mine.ensure((size_t) vmax * 16 * 8) || !all.ensure((size_t) vmax * 16 * 8 * (size_t) world)) return fail("device buffers of the voxel merge");
        if (hipMemset(mine.p, 0, (size_t) vmax * 16 * 8) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return fail("hipMemset");  // rows with n == 0 are padding
        int64_t got = 0;
        DIST_TRY(me_voxel_partial_rows_device(ctx_, slot, param_.vmd_voxel_size_, mine.as<double>(), vmax, &got));
        COMM_TRY(comm_->all_gather(mine.p, all.p, (size_t) vmax * 16 * 8));
        DIST_TRY(me_voxel_merge_device(ctx_, slot, param_.vmd_voxel_size_, all.as<double>(), vmax * world));
        if (param_.enable_debug)
            std::cerr << "[rank " << rank << "] voxel partials of slot " << slot << ": " << got << " rows here, padded to " << vmax << " per rank" << std::endl;
    }
    calculateVMD(/*tables_ready=*/true, /*write_files=*/rank == 0);
    if (!last_error.empty()) return -1;
    mark("voxel_awd_scs");
    if (rank == 0 && param_.save_immediate_result_) {
        std::swap(ctx_, render_ctx_);  // me_render_distance on the whole-cloud context
        saveRegistrationResults();
        std::swap(ctx_, render_ctx_);
        mark("result_files");
    }
    if (rank == 0) {
        std::cout << "INFO: multi-GPU phases on rank 0 [ms]:";
        for (const auto &ph : phases) std::cout << " " << ph.first << "=" << std::fixed << std::setprecision(1) << ph.second;
        std::cout << " total=" << phase_clock.toc() << std::defaultfloat << std::setprecision(6) << std::endl;
    }
    return last_error.empty() ? 0 : -1;
}
