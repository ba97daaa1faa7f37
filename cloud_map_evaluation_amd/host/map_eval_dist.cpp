// map_eval_dist.cpp — `num_gpus: N` for the drop-in binary: MapEval::process() between "clouds loaded and down-sampled"
// and "results written" (map_eval/src/map_eval.cpp:51-85) sharded over N GPUs, one process per GPU, collectives over RCCL
// (dist_comm.hpp).  C++ restatement of cloud_map_evaluation_amd/dist.py::suite_step_slab on the library's slab entry points
// (include/mapeval_hip.h: me_set_slab, me_nn_unresolved / me_nn_points_bounded / me_nn_patch, me_nn_partial_sums /
// me_nn_sigma_sums / me_nn_finalize, me_voxel_partial_rows_device / me_voxel_merge_device, me_slab_points):
//
//   every rank reads both files, cuts the ground truth's longest axis into N equal-count slabs (same data, same
//   arithmetic -> same cuts, no collective), keeps slab + halo of both clouds (me_set_slab + me_upload_cloud), then
//     MME      per-point on the slab, {sum H, n_valid} all-reduced                        (1 collective, 4 doubles)
//     AC/COM   local 1-NN both ways; queries that a closer point on another rank could beat go through the cross-rank
//              step: counts all-gather, queries + bounds all-gather, bounded search on every rank, MIN all-reduce, patch
//              (3 collectives); partial sums all-reduce, sigma numerators all-reduce          (2 collectives)
//     AWD/SCS  voxel partial rows of the owned points: counts ride on the sums, one padded all-gather per cloud, Chan
//              merge on the device, me_awd_scs on the merged (replicated) tables               (2 collectives)
//     outputs  per-point entropies / squared distances of the owned points are scattered into whole-cloud arrays and
//              summed over the ranks; rank 0 hands them to a second, whole-cloud context (me_set_mme_result /
//              me_set_nn_result) and writes every file exactly as the single-GPU path does.
//   The ICP path (evaluate_using_initial: false) stays single-GPU.  With a non-identity initial_matrix the map is transformed
//   BEFORE the MME pass (the reference transforms it after, :1206): for a rigid matrix the entropies agree to rounding.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>

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
    medist::DevMem buf;
    if (!buf.ensure(v.size() * 8) || !h2d(buf.p, v.data(), v.size() * 8)) return fail("device staging buffer");
    COMM_TRY(min_op ? comm_->all_reduce_min_f64(buf.as<double>(), v.size()) : comm_->all_reduce_sum_f64(buf.as<double>(), v.size()));
    if (!d2h(v.data(), buf.p, v.size() * 8)) return fail("device staging buffer");
    return 0;
}

// owned entries of a per-point array of this rank's slab -> the whole-cloud array (every point is owned by exactly one rank:
// the sum over the ranks IS the array)
int MapEval::gatherPerPoint(int slot, size_t n_global, const std::vector<double> *vals, const std::vector<uint8_t> *flags,
                            std::vector<double> *vals_out, std::vector<uint8_t> *flags_out) {
    int64_t n_loc = 0;
    DIST_TRY(me_slab_points(ctx_, slot, nullptr, nullptr, 0, &n_loc));
    std::vector<int64_t> orig((size_t) n_loc);
    std::vector<uint8_t> owned((size_t) n_loc);
    if (n_loc) DIST_TRY(me_slab_points(ctx_, slot, orig.data(), owned.data(), n_loc, &n_loc));
    medist::DevMem buf;
    if (vals) {
        vals_out->assign(n_global, 0.0);
        for (int64_t i = 0; i < n_loc; ++i)
            if (owned[(size_t) i]) (*vals_out)[(size_t) orig[(size_t) i]] = (*vals)[(size_t) i];
        if (!buf.ensure(n_global * 8) || !h2d(buf.p, vals_out->data(), n_global * 8)) return fail("device staging buffer");
        COMM_TRY(comm_->all_reduce_sum_f64(buf.as<double>(), n_global));
        if (!d2h(vals_out->data(), buf.p, n_global * 8)) return fail("device staging buffer");
    }
    if (flags) {
        flags_out->assign(n_global, 0);
        for (int64_t i = 0; i < n_loc; ++i)
            if (owned[(size_t) i]) (*flags_out)[(size_t) orig[(size_t) i]] = (*flags)[(size_t) i];
        if (!buf.ensure(n_global) || !h2d(buf.p, flags_out->data(), n_global)) return fail("device staging buffer");
        COMM_TRY(comm_->all_reduce_sum_u8(buf.as<uint8_t>(), n_global));
        if (!d2h(flags_out->data(), buf.p, n_global)) return fail("device staging buffer");
    }
    return 0;
}

int MapEval::processDist(double t_loaded) {
    const int rank = comm_->rank, world = comm_->world;
    const size_t n_e = map_3d_->size(), n_g = gt_3d_->size();
    if (!param_.evaluate_using_initial_)
        return fail("num_gpus > 1 runs the initial-matrix path; ICP registration (evaluate_using_initial: false) is single-GPU");
    t1 = t_loaded;
    // *map_3d_ = map_3d_->Transform(initial_matrix) (:1206) on the whole cloud, before it is cut into slabs
    bool identity = true;
    for (int i = 0; i < 16; ++i) identity = identity && (param_.initial_matrix_[i] == ((i % 5 == 0) ? 1.0 : 0.0));
    if (!identity) {
        DIST_TRY(me_transform_cloud(ctx_, ME_SLOT_EST, param_.initial_matrix_.data()));
        DIST_TRY(me_download_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data()));
    }
    // ---- slab faces: equal-count cuts along the longest axis of the ground truth (identical on every rank) ----
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n_g; ++i)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], gt_3d_->points_[3 * i + d]);
            hi[d] = std::max(hi[d], gt_3d_->points_[3 * i + d]);
        }
    int axis = 0;
    for (int d = 1; d < 3; ++d)
        if (hi[d] - lo[d] > hi[axis] - lo[axis]) axis = d;
    std::vector<double> cuts((size_t) world + 1);
    {
        std::vector<double> coord(n_g);
        for (size_t i = 0; i < n_g; ++i) coord[i] = gt_3d_->points_[3 * i + axis];
        cuts[0] = -INFINITY;
        cuts[(size_t) world] = INFINITY;
        for (int k = 1; k < world; ++k) {
            const size_t at = (size_t) ((unsigned long long) n_g * (unsigned long long) k / (unsigned long long) world);
            std::nth_element(coord.begin(), coord.begin() + (std::ptrdiff_t) at, coord.end());
            cuts[(size_t) k] = coord[at];
            if (!(cuts[(size_t) k] > cuts[(size_t) k - 1])) cuts[(size_t) k] = std::nextafter(cuts[(size_t) k - 1], INFINITY);
        }
    }
    const double halo = std::max(1.0, 1.0001 * param_.nn_radius_);  // MME needs halo >= nn_radius; 1 m covers the usual 1-NN reach
    DIST_TRY(me_set_slab(ctx_, axis, cuts[(size_t) rank], cuts[(size_t) rank + 1], halo));
    DIST_TRY(me_upload_cloud(ctx_, ME_SLOT_EST, map_3d_->points_.data(), (int64_t) n_e, nullptr, param_.nn_radius_));
    DIST_TRY(me_upload_cloud(ctx_, ME_SLOT_GT, gt_3d_->points_.data(), (int64_t) n_g, nullptr, param_.nn_radius_));
    if (rank == 0)
        std::cout << "INFO: multi-GPU run: " << world << " rank(s) over " << comm_->name() << ", slabs along axis " << axis << ", halo "
                  << halo << " m" << std::endl;
    // rank 0 keeps a second context with the WHOLE clouds for the colour renderers (only when files are written)
    if (rank == 0 && param_.save_immediate_result_) {
        render_ctx_ = me_create(param_.gpu_device, 0);
        if (!render_ctx_) return fail("GPU engine unavailable for the renderers");
        if (me_upload_cloud(render_ctx_, ME_SLOT_EST, map_3d_->points_.data(), (int64_t) n_e, nullptr, param_.nn_radius_) != ME_OK ||
            me_upload_cloud(render_ctx_, ME_SLOT_GT, gt_3d_->points_.data(), (int64_t) n_g, nullptr, param_.nn_radius_) != ME_OK)
            return fail(me_last_error(render_ctx_));
    }
    TicToc3 clock;

    // ---- MME (computeMME, :149-189): est k >= 10, gt k >= 5 ----
    std::vector<double> sums(4, 0.0);
    if (param_.evaluate_mme_) {
        for (int pass = 0; pass < (param_.evaluate_gt_mme_ ? 2 : 1); ++pass) {
            const int slot = pass == 0 ? ME_SLOT_EST : ME_SLOT_GT;
            const int64_t n_loc = me_cloud_size(ctx_, slot);
            std::vector<double> ent((size_t) n_loc, 0.0);
            std::vector<uint8_t> val((size_t) n_loc, 0);
            double s = 0;
            int64_t nv = 0;
            DIST_TRY(me_mme(ctx_, slot, param_.nn_radius_, pass == 0 ? 10 : 5, ent.data(), val.data(), &s, &nv));
            sums[(size_t) 2 * pass] = s;
            sums[(size_t) 2 * pass + 1] = (double) nv;
            std::vector<uint8_t> gv;
            if (gatherPerPoint(slot, pass == 0 ? n_e : n_g, &ent, &val, pass == 0 ? &est_entropies : &gt_entropies,
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

    // ---- AC / COM / CD (calculateMetricsWithInitialMatrix, :1204-1260) ----
    const int dirs[2][2] = {{ME_SLOT_EST, ME_SLOT_GT}, {ME_SLOT_GT, ME_SLOT_EST}};
    int64_t cnt[2] = {0, 0};
    for (int d = 0; d < 2; ++d) {
        DIST_TRY(me_nn1(ctx_, dirs[d][0], dirs[d][1], nullptr, nullptr));
        DIST_TRY(me_nn_unresolved(ctx_, dirs[d][0], nullptr, nullptr, 0, &cnt[d]));
    }
    {   // cross-rank step: open queries of both directions in one all-gather, their answers in one MIN-reduce
        std::vector<double> table((size_t) world * 2, 0.0);
        table[(size_t) rank * 2] = (double) cnt[0];
        table[(size_t) rank * 2 + 1] = (double) cnt[1];
        if (allReduceHost(table, false) != 0) return -1;
        int64_t cmax[2] = {0, 0}, total = 0;
        for (int k = 0; k < world; ++k)
            for (int d = 0; d < 2; ++d) {
                cmax[d] = std::max<int64_t>(cmax[d], (int64_t) table[(size_t) k * 2 + d]);
                total += (int64_t) table[(size_t) k * 2 + d];
            }
        if (total > 0 && world > 1) {
            // message of a rank: [X0 (cmax0 x 3) | D0 (cmax0) | X1 (cmax1 x 3) | D1 (cmax1)] doubles
            const size_t msg_len = (size_t) (cmax[0] + cmax[1]) * 4;
            const size_t off_x[2] = {0, (size_t) cmax[0] * 4}, off_d[2] = {(size_t) cmax[0] * 3, (size_t) cmax[0] * 4 + (size_t) cmax[1] * 3};
            medist::DevMem msg, all, ans;
            if (!msg.ensure(msg_len * 8) || !all.ensure(msg_len * 8 * (size_t) world) || !ans.ensure((size_t) (cmax[0] + cmax[1]) * 8 * (size_t) world))
                return fail("device buffers of the cross-rank step");
            // (hipMemset / device-to-device hipMemcpy may return before they are done, and the library works on its own
            //  non-blocking stream: settle them before handing the buffer over)
            if (hipMemset(msg.p, 0, msg_len * 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail("hipMemset");
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
                    if (hipMemcpy(dst, src + off_d[d], (size_t) c * 8, hipMemcpyDeviceToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
                        return fail("hipMemcpy");
                    if (k != rank) DIST_TRY(me_nn_points_bounded(ctx_, dirs[d][1], src + off_x[d], c, dst));
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
        DIST_TRY(me_nn_partial_sums(ctx_, dirs[d][0], param_.icp_max_distance_, ME_GATE_LE_UNSQUARED, param_.trunc_dist_.data(), &p));
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
        DIST_TRY(me_nn_sigma_sums(ctx_, dirs[d][0], param_.icp_max_distance_, ME_GATE_LE_UNSQUARED, mean, sig.data() + 5 * d));
    }
    if (allReduceHost(sig, false) != 0) return -1;
    me_nn_stats_out eg, ge;
    me_nn_finalize(&tot[0], sig.data(), (int64_t) n_e, &eg);
    me_nn_finalize(&tot[1], sig.data() + 5, (int64_t) n_g, &ge);
    finishInitialMatrixMetrics(eg, ge, clock.toc() / 1000.0);
    // squared distances of the map's points for raw_rendered_dis_map.pcd / inlier_rendered_dis_map.pcd (:485-495)
    if (param_.save_immediate_result_) {
        const int64_t n_loc = me_cloud_size(ctx_, ME_SLOT_EST);
        std::vector<double> d2((size_t) n_loc), d2_all;
        if (n_loc) DIST_TRY(me_nn_fetch(ctx_, ME_SLOT_EST, nullptr, d2.data()));
        if (gatherPerPoint(ME_SLOT_EST, n_e, &d2, nullptr, &d2_all, nullptr) != 0) return -1;
        if (render_ctx_ && me_set_nn_result(render_ctx_, ME_SLOT_EST, ME_SLOT_GT, d2_all.data()) != ME_OK) return fail(me_last_error(render_ctx_));
    }
    t5 = t4 = t3 = t1 + clock.toc();

    // ---- AWD / CDF / SCS (calculateVMD, :240-390): Chan merge of every rank's voxel partials, then the replicated tables ----
    for (int s = 0; s < 2; ++s) {
        const int slot = s == 0 ? ME_SLOT_EST : ME_SLOT_GT;
        int64_t vmax = 1;
        for (int k = 0; k < world; ++k) vmax = std::max<int64_t>(vmax, (int64_t) std::llround(vec[(size_t) 2 * kPart + (size_t) s * world + (size_t) k]));
        medist::DevMem mine, all;
        if (!mine.ensure((size_t) vmax * 16 * 8) || !all.ensure((size_t) vmax * 16 * 8 * (size_t) world)) return fail("device buffers of the voxel merge");
        if (hipMemset(mine.p, 0, (size_t) vmax * 16 * 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail("hipMemset");  // rows with n == 0 are padding
        int64_t got = 0;
        DIST_TRY(me_voxel_partial_rows_device(ctx_, slot, param_.vmd_voxel_size_, mine.as<double>(), vmax, &got));
        COMM_TRY(comm_->all_gather(mine.p, all.p, (size_t) vmax * 16 * 8));
        DIST_TRY(me_voxel_merge_device(ctx_, slot, param_.vmd_voxel_size_, all.as<double>(), vmax * world));
        if (param_.enable_debug)
            std::cerr << "[rank " << rank << "] voxel partials of slot " << slot << ": " << got << " rows here, padded to " << vmax << " per rank" << std::endl;
    }
    calculateVMD(/*tables_ready=*/true, /*write_files=*/rank == 0);
    if (!last_error.empty()) return -1;
    if (rank == 0 && param_.save_immediate_result_) {
        std::swap(ctx_, render_ctx_);  // me_render_distance on the whole-cloud context
        saveRegistrationResults();
        std::swap(ctx_, render_ctx_);
    }
    return last_error.empty() ? 0 : -1;
}
