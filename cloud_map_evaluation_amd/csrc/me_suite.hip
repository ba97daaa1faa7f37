// me_suite.hip — the whole hot path in ONE call: what MapEval::process() runs between "clouds loaded" and "results written"
// (map_eval.cpp:52-85: computeMME :56, calculateMetricsWithInitialMatrix :76, calculateVMD :85).
//
// me_run_suite        the stages back to back on the context's stream, clouds already uploaded (round 1).
// me_run_suite_from   the same stages starting from the two RAW clouds (host or device pointers), scheduled on TWO lanes: the
//                     calling thread drives `ctx`, an internal host thread drives me_twin(ctx) — the schedule that bench.py
//                     measured from Python through round 4 (dist._Lane), now behind the C ABI so that a C++ caller (the bundled
//                     host, the INTEGRATION.md section B patch of the reference's own process()) gets it with one call from its
//                     one thread.  Same kernels, same call order per product, results bit-identical to the sequential call.
//
//   main lane (ctx, highest stream priority)            second lane (twin, lowest priority)
//   ---------------------------------------            ----------------------------------------------------------------
//   upload + index the map, WITHOUT its octree          [host input: wait until the map has crossed the link]
//   (the gathers also emit the voxel run records        upload + index the ground truth, without its octree (its radix sort
//    for vmd_voxel_size: me_vox_rows.hpp)                BEFORE the map's MME starts: onesweep crawls beside a full chip; the
//   wait: ground truth's sort queued (event)             rest under that MME)
//   MME of the map            (VALU-bound)              -> "ground truth indexed"; its octree (cloud_finish_octree)
//   [T: transform + re-index the map, :1206]            voxel Gaussians of the ground truth, then of the map, from the gathers'
//   the map's octree (cloud_finish_octree)               records (~40 launch-sized kernels each: they hide under the MMEs)
//   [host input: voxel Gaussians of the map]            wait: map's octree
//   wait: ground truth indexed                          1-NN ground truth -> map + partial sums + ITS sigma pass
//   MME of the ground truth                             [no records (three-pass build): the voxel tables here, after the search;
//   wait: ground truth's octree                          the map's by whichever lane gets to it first]
//   1-NN map -> ground truth + partial sums + ITS sigma pass
//   AWD / CDF / SCS if both tables are ready (beside the second lane's tail); join; [AWD / CDF / SCS otherwise]
//
// Measured on the 50 M + 50 M bench pair (profiles/EXPERIMENTS.md "Round 6"): the step is the SUM of its kernels' work on the vector
// unit whatever the order — what the order decides is how much launch-latency-bound work (octrees, voxel tables, sigma, AWD) is left
// standing alone at either end of the step.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <thread>

#include "me_internal.hpp"

namespace {

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

bool is_identity(const double *T) {
    if (!T) return true;
    for (int i = 0; i < 16; ++i)
        if (T[i] != ((i % 5 == 0) ? 1.0 : 0.0)) return false;
    return true;
}

// the second lane of me_run_suite_from
struct SuiteLane {
    me_ctx *t = nullptr;  // the twin context
    const me_suite_params *p = nullptr;
    const double *gt = nullptr;
    long long n_gt = 0;
    bool gt_on_device = false, upload_gt = false, wait_for_link = false, pin_gt = false;
    bool est_voxel_on_main = false;  // host input: the main lane builds the map's voxel table while the ground truth is in flight
    std::atomic<int> est_voxel_taken{0};  // device input: the lane that sets it builds the map's voxel table
    bool gt_pinned = false;  // this call page-locked the ground truth's buffer (released by the caller of run())
    std::mutex m;
    std::condition_variable cv;
    bool est_on_device = false;  // main -> lane: the map's H2D copy is over (the link is free)
    bool est_final = false;      // main -> lane: the map is indexed in its final pose
    bool gt_ready = false;       // lane -> main: the ground truth is indexed (or the lane has failed)
    bool aborted = false;        // main -> lane: stop at the next wait
    std::atomic<int> rc{ME_OK};
    bool finished = false;       // lane -> main: run() has returned (the lane touches nothing of this object afterwards)
    bool started = false;        // the worker has been handed this lane
    me_nn_partial back{};        // ground truth -> map partial sums
    double back_sig[5] = {0, 0, 0, 0, 0};  // ... and the sigma numerators of its second pass
    bool est_tree = false;       // main -> lane: the map's octree is complete (it is built after the map's MME)
    bool gt_tree = false;        // lane -> main: the ground truth's octree is complete (built after gt_ready)
    bool tables_ready = false;   // lane -> main: both voxel tables are complete (built from the gather's records, before the search)
    // Round 6: rocPRIM's onesweep radix pass uses decoupled lookback, and under a chip filled by the main lane's k_mme3 ONE pass of the
    // ground truth's sort took 12.8 ms instead of 0.3 (every other kernel of its index build ran at its normal speed beside the MME:
    // profiles/r06_timeline.txt) — the ground truth was indexed at 21.4 ms, the main lane waited 1 ms for it, and this lane's search
    // could not start before.  The lane records an event when its sort has been queued; the main lane makes its stream wait for that
    // event before the map's MME: ~0.3 ms later for the MME, ~12 ms earlier for everything on this lane.
    hipEvent_t sort_event = nullptr;
    bool sort_queued = false;    // lane -> main: sort_event has been recorded (or there is no sort to wait for)
    static void sort_hook(void *self, hipStream_t stream) {
        SuiteLane *l = static_cast<SuiteLane *>(self);
        if (l->sort_event && !l->sort_queued) {
            (void) hipEventRecord(l->sort_event, stream);
            l->set(&SuiteLane::sort_queued);
        }
    }

    void set(bool SuiteLane::*flag) {
        {
            std::lock_guard<std::mutex> g(m);
            this->*flag = true;
        }
        cv.notify_all();
    }
    bool is_set(bool SuiteLane::*flag) {
        std::lock_guard<std::mutex> g(m);
        return this->*flag;
    }
    // false: aborted
    bool wait(bool SuiteLane::*flag) {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return this->*flag || aborted; });
        return !aborted;
    }
    void run() {
        int r;
        try {  // (nothing may unwind out of the worker thread, nor across the C boundary)
            r = body();
        } catch (const std::bad_alloc &) {
            r = t->fail(ME_ERR_HIP, "me_run_suite_from: out of host memory on the second lane");
        } catch (const std::exception &e) {
            r = t->fail(ME_ERR_HIP, std::string("me_run_suite_from: second lane: ") + e.what());
        }
        rc = r;
        {   // gt_ready: a failed lane must not leave the main lane waiting; finished: last touch of this object
            std::lock_guard<std::mutex> g(m);
            gt_ready = true;
            gt_tree = true;
            sort_queued = true;
            finished = true;
        }
        cv.notify_all();
    }
    int body() {
        if (hipSetDevice(t->device) != hipSuccess) return t->fail(ME_ERR_HIP, "me_run_suite_from: hipSetDevice failed on the second lane");
        if (upload_gt) {
            // clouds that start in HOST memory share the PCIe link: one after the other, the ground truth crosses it under the
            // map's MME kernel.  Device-resident clouds: both lanes start at once.
            if (pin_gt) {
                gt_pinned = hipHostRegister(const_cast<double *>(gt), (size_t) n_gt * 24, hipHostRegisterDefault) == hipSuccess;
                // (a refused registration — the buffer is pinned already — is not an error of the call; hipGetLastError is sticky PER
                // THREAD on ROCm 7: left pending it would surface in this lane's next ME_CHECK(hipGetLastError()))
                (void) hipGetLastError();
            }
            if (wait_for_link && !wait(&SuiteLane::est_on_device)) return ME_OK;
            t->sort_hook = sort_event ? &SuiteLane::sort_hook : nullptr;
            t->sort_hook_arg = this;
            const int urc = me::cloud_upload(t, ME_SLOT_GT, gt, gt_on_device, n_gt, nullptr, p->nn_radius);
            t->sort_hook = nullptr;
            if (urc != ME_OK) return urc;
        }
        set(&SuiteLane::sort_queued);  // (nothing to wait for any more, whatever happened above)
        set(&SuiteLane::gt_ready);
        // (the main lane's MME of the ground truth starts on the cell tables; the octree is for the tails of the searches)
        ME_TRY(me::cloud_finish_octree(t, ME_SLOT_GT));
        set(&SuiteLane::gt_tree);
#if ME_TUNE_SUITE_NN_FIRST
        // Round 6: the reverse search before the voxel PASSES.  The search is VALU-bound like the main lane's MME of the ground
        // truth beside it — together they keep the vector unit busy —, and three-pass voxel tables (HBM-bound, 2.7 ms alone) then run
        // under the main lane's own search.  In the other order the low-priority voxel passes starved under the MME (19 ms for 2.7 ms
        // of work), the reverse search started when the main lane's had finished, and for ~3 ms in between only HBM-bound kernels and
        // the octree tails were running (profiles/r05_timeline_two_lane.txt).
        // A table whose run records the gather has already emitted (k_gather<VOX>) is another matter: what is left of it is ~40 launches
        // of a few microseconds each (compaction, a merge sort of ~n / 60 records, one reduction) and three mailbox reads — 0.1 ms of
        // work, 0.4 ms of latency.  After the searches that chain was the END of the step, on an idle chip (profiles/r06_timeline.txt:
        // 43.0 -> 43.8 ms); here it hides under the MME kernels.
        bool gt_table = false;
#if ME_TUNE_SUITE_VOX_EARLY
        if (t->cloud[ME_SLOT_GT].vox_rec_valid) {
            ME_TRY(me::voxel_build(t, ME_SLOT_GT, p->vmd_voxel_size, false));
            gt_table = true;
        }
#endif
        if (!wait(&SuiteLane::est_final)) return ME_OK;
#if ME_TUNE_SUITE_VOX_EARLY
        if (gt_table && !est_voxel_on_main && t->cloud[ME_SLOT_EST].vox_rec_valid && !est_voxel_taken.exchange(1)) {
            ME_TRY(me::voxel_build(t, ME_SLOT_EST, p->vmd_voxel_size, false));
            set(&SuiteLane::tables_ready);  // (both tables complete: voxel_build returns after its stream has drained)
        }
#endif
        if (!wait(&SuiteLane::est_tree)) return ME_OK;
        ME_TRY(me::nn_search(t, ME_SLOT_GT, ME_SLOT_EST));
        ME_TRY(me::nn_partial(t, ME_SLOT_GT, p->icp_max_distance, p->gate_mode, p->trunc, &back));
        ME_TRY(back_sigma());
        if (!gt_table) ME_TRY(me::voxel_build(t, ME_SLOT_GT, p->vmd_voxel_size, false));
        // the map's voxel table: whichever lane gets to it first (never both: same buffers) — the main lane claims it when its own
        // search is over and this lane is still busy with the ground truth's table
        if (!est_voxel_on_main && !est_voxel_taken.exchange(1)) ME_TRY(me::voxel_build(t, ME_SLOT_EST, p->vmd_voxel_size, false));
#else
        ME_TRY(me::voxel_build(t, ME_SLOT_GT, p->vmd_voxel_size, false));
        if (!wait(&SuiteLane::est_final)) return ME_OK;
        if (!est_voxel_on_main) ME_TRY(me::voxel_build(t, ME_SLOT_EST, p->vmd_voxel_size, false));  // (never both lanes: same buffers)
        if (!wait(&SuiteLane::est_tree)) return ME_OK;
        ME_TRY(me::nn_search(t, ME_SLOT_GT, ME_SLOT_EST));
        ME_TRY(me::nn_partial(t, ME_SLOT_GT, p->icp_max_distance, p->gate_mode, p->trunc, &back));
        ME_TRY(back_sigma());
#endif
        return ME_OK;
    }
    // second pass of the ground truth -> map statistics (map_eval.cpp:1132-1138) on THIS lane, as soon as its means exist: the main
    // lane's own second pass and AWD / SCS run beside it instead of after it
    int back_sigma() {
        double mean[5];
        for (int k = 0; k < 5; ++k) mean[k] = back.sum_d[k] / (double) back.n_corr;
        return me::nn_sigma(t, ME_SLOT_GT, p->icp_max_distance, p->gate_mode, mean, back_sig);
    }
    // waits until run() has returned; false: the lane failed
    bool join() {
        if (started) {
            std::unique_lock<std::mutex> g(m);
            cv.wait(g, [&] { return finished; });
        }
        return rc.load() == ME_OK;
    }
    void abort_and_join() {
        {
            std::lock_guard<std::mutex> g(m);
            aborted = true;
        }
        cv.notify_all();
        (void) join();
    }
    ~SuiteLane() { abort_and_join(); }  // (the worker must be done with this object before it dies, whatever path leaves the call)
};

// The second lane's host thread lives in the CONTEXT, not in the call (round 6): it is created by the first overlapped
// me_run_suite_from, sleeps on a condition variable between calls and is joined by me_destroy.  A std::thread per call cost the step
// its creation, the new thread's first hipSetDevice and its exit (~0.1 - 0.2 ms on the critical path of a 47 ms step, and on some
// boxes more: VERDICT round 5, weak 3).
struct LaneWorker {
    std::mutex m;
    std::condition_variable cv;
    SuiteLane *job = nullptr;
    bool quit = false;
    std::thread th;
    void loop() {
        for (;;) {
            SuiteLane *j = nullptr;
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return job != nullptr || quit; });
                if (quit) return;
                j = job;
                job = nullptr;
            }
            j->run();
        }
    }
    void post(SuiteLane *j) {
        {
            std::lock_guard<std::mutex> g(m);
            job = j;
        }
        j->started = true;
        cv.notify_all();
    }
    ~LaneWorker() {
        {
            std::lock_guard<std::mutex> g(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
void free_lane_worker(void *w) { delete static_cast<LaneWorker *>(w); }

// the context's lane worker, created on first use; nullptr: the thread could not be created (message set)
LaneWorker *lane_worker(me_ctx *ctx) {
    if (ctx->suite_worker) return static_cast<LaneWorker *>(ctx->suite_worker);
    try {
        LaneWorker *w = new LaneWorker();
        try {
            w->th = std::thread([w] { w->loop(); });
        } catch (...) {
            delete w;
            throw;
        }
        ctx->suite_worker = w;
        ctx->suite_worker_free = &free_lane_worker;
        return w;
    } catch (const std::exception &e) {
        ctx->fail(ME_ERR_HIP, std::string("me_run_suite_from: cannot start the second lane's thread: ") + e.what());
        return nullptr;
    }
}

// mme_run rebuilds a slot's index when its cells do not fit the radius (me_mme.hip): the same predicate, so that the rebuild
// can be done BEFORE two lanes share the cloud
bool index_fits_radius(const me::Cloud &c, double radius) {
    if (!(radius > 0)) return c.index_valid;  // (no radius given: any index serves the 1-NN and voxel stages)
    const double want_h = radius * (1.0 + 0x1p-20);
    return c.index_valid && c.cell_h >= want_h && c.cell_h <= 1.5 * want_h;
}

// second pass of one direction: sigma needs the mean of every threshold (map_eval.cpp:1132-1138)
int sigma_pass(me_ctx *ctx, int slot, const me_suite_params *p, const me_nn_partial &part, double *sig) {
    double mean[5];
    for (int k = 0; k < 5; ++k) mean[k] = part.sum_d[k] / (double) part.n_corr;
    return me::nn_sigma(ctx, slot, p->icp_max_distance, p->gate_mode, mean, sig);
}

}  // namespace

extern "C" {

int me_run_suite(me_ctx *ctx, const me_suite_params *p, me_suite_out *out) {
    if (!ctx) return ME_ERR_ARG;
    if (!p || !out) return ctx->fail(ME_ERR_ARG, "me_run_suite: NULL argument");
    if (ctx->shard_world != 1 || ctx->slab.axis >= 0) return ctx->fail(ME_ERR_STATE, "me_run_suite is single-GPU; drive the partial calls when sharded");
    std::memset(out, 0, sizeof(*out));
    const auto t_all = Clock::now();
    // stage_ms: host wall clock per stage (every stage below ends with a stream synchronisation)
    // MME first, as MapEval::process (map_eval.cpp:52-66)
    if (p->evaluate_mme) {
        double s = 0;
        int64_t nv = 0;
        auto t0 = Clock::now();
        ME_TRY(me_mme(ctx, ME_SLOT_EST, p->nn_radius, 10, nullptr, nullptr, &s, &nv));  // k >= 10 (:1675)
        out->stage_ms[4] = ms_since(t0);
        out->mme_est = nv > 0 ? s / (double) nv : 0.0;
        out->mme_est_valid = nv;
        if (p->evaluate_gt_mme) {
            t0 = Clock::now();
            ME_TRY(me_mme(ctx, ME_SLOT_GT, p->nn_radius, 5, nullptr, nullptr, &s, &nv));  // k >= 5 (:1458)
            out->stage_ms[5] = ms_since(t0);
            out->mme_gt = nv > 0 ? s / (double) nv : 0.0;
            out->mme_gt_valid = nv;
        }
    }
    // AC / COM both directions (:1213-1242) + full CD (:1398-1431) from the same two searches
    auto t0 = Clock::now();
    ME_TRY(me::nn_search(ctx, ME_SLOT_EST, ME_SLOT_GT));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    out->stage_ms[1] = ms_since(t0);
    t0 = Clock::now();
    ME_TRY(me_nn_stats(ctx, ME_SLOT_EST, p->icp_max_distance, p->gate_mode, p->trunc, &out->est_gt));
    out->stage_ms[3] = ms_since(t0);
    t0 = Clock::now();
    ME_TRY(me::nn_search(ctx, ME_SLOT_GT, ME_SLOT_EST));
    ME_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    out->stage_ms[2] = ms_since(t0);
    t0 = Clock::now();
    ME_TRY(me_nn_stats(ctx, ME_SLOT_GT, p->icp_max_distance, p->gate_mode, p->trunc, &out->gt_est));
    out->stage_ms[3] += ms_since(t0);
    out->full_chamfer = out->est_gt.mean_nn_dist + out->gt_est.mean_nn_dist;
    // AWD / SCS (:85, :240-390)
    int64_t n_rows = 0;
    t0 = Clock::now();
    ME_TRY(me_awd_scs(ctx, p->vmd_voxel_size, p->min_pts > 0 ? p->min_pts : 100, p->scs_radius > 0 ? p->scs_radius : 5, nullptr,
                      nullptr, &n_rows, &out->awd, &out->scs, nullptr));
    out->stage_ms[6] = ms_since(t0);
    out->n_w_voxels = n_rows;
    out->stage_ms[7] = ms_since(t_all);
    return ME_OK;
}

int me_run_suite_from(me_ctx *ctx, const double *est, int64_t n_est, const double *gt, int64_t n_gt, const double *T,
                      const me_suite_params *p, int flags, me_suite_out *out) {
    if (!ctx) return ME_ERR_ARG;
    if (!p || !out) return ctx->fail(ME_ERR_ARG, "me_run_suite_from: NULL argument");
    if (ctx->is_twin) return ctx->fail(ME_ERR_ARG, "me_run_suite_from: call it on the primary context");
    if (ctx->shard_world != 1 || ctx->slab.axis >= 0)
        return ctx->fail(ME_ERR_STATE, "me_run_suite_from is single-GPU; drive the partial calls when sharded");
    if ((est == nullptr) != (gt == nullptr)) return ctx->fail(ME_ERR_ARG, "me_run_suite_from: pass both clouds, or neither (clouds already uploaded)");
    const bool upload = est != nullptr;
    if (!upload && (!ctx->cloud[ME_SLOT_EST].uploaded || !ctx->cloud[ME_SLOT_GT].uploaded))
        return ctx->fail(ME_ERR_STATE, "me_run_suite_from: no clouds passed and none uploaded");
    const bool on_device = (flags & ME_SUITE_DEVICE_INPUT) != 0;
    const bool overlap = (flags & ME_SUITE_OVERLAP) != 0;
    const bool pin = upload && !on_device && (flags & ME_SUITE_PIN_HOST_INPUT) != 0;
    const bool moved = !is_identity(T);  // the map is evaluated for MME as loaded and transformed afterwards (:56 then :1206)
    std::memset(out, 0, sizeof(*out));
    ME_CHECK(ctx, hipSetDevice(ctx->device));
    const auto t_all = Clock::now();
    // the index builds of this call also emit the voxel run records for vmd_voxel_size (me_index.hip: k_gather<VOX>), on both lanes
    // ... and leave the octrees to cloud_finish_octree when an MME pass comes first (it needs the cell tables only)
    struct VoxHint {  // (what me_set_voxel_hint had set comes back when the call is over)
        me_ctx *a, *b = nullptr;
        bool defer;
        double keep_a, keep_b = 0;
        VoxHint(me_ctx *c, double v, bool d) : a(c), defer(d), keep_a(c->vox_hint) {
            a->vox_hint = v;
            a->defer_octree = d;
        }
        void also(me_ctx *t, double v) {
            b = t;
            if (b) {
                keep_b = b->vox_hint;
                b->vox_hint = v;
                b->defer_octree = defer;
            }
        }
        ~VoxHint() {
            a->vox_hint = keep_a;
            a->defer_octree = false;
            if (b) {
                b->vox_hint = keep_b;
                b->defer_octree = false;
            }
        }
    } vox_hint(ctx, upload ? p->vmd_voxel_size : 0.0, ME_TUNE_SUITE_DEFER_OCTREE && upload && p->evaluate_mme);

    if (!upload) {
        // Resident clouds: with two lanes both start at once on the SAME two Cloud objects, and a stage that finds a
        // slot's index missing or unfit rebuilds it — mme_run when the cells do not fit the radius (the documented default of an upload
        // with cell_size <= 0), voxel_build / nn_search when there is none — re-sorting `sp` and reallocating the tables under the
        // other lane's kernels.  Every index is therefore settled here, on one lane, before the second one starts: on the lattice of
        // nn_radius, which is what the call builds from raw clouds — the same sorted order, so the same sums bit for bit (and the
        // 1-NN grid is chosen among the levels below the cell: under 1.5 m "automatic" cells it would hold hundreds of points).
        for (int slot = 0; slot < 2; ++slot) {
            const me::Cloud &c = ctx->cloud[slot];
            if (c.n > 0 && !index_fits_radius(c, p->nn_radius)) ME_TRY(me::cloud_build_index(ctx, slot, p->nn_radius));
        }
    }
    SuiteLane lane;
    LaneWorker *worker = nullptr;
    if (overlap) {
        lane.t = me_twin(ctx);
        if (!lane.t) return ME_ERR_HIP;  // (me_twin has set the message)
        vox_hint.also(lane.t, upload ? p->vmd_voxel_size : 0.0);
        worker = lane_worker(ctx);
        if (!worker) return ME_ERR_HIP;
        lane.p = p;
        lane.gt = gt;
        lane.n_gt = n_gt;
        lane.gt_on_device = on_device;
        lane.upload_gt = upload;
        lane.wait_for_link = upload && !on_device;
        lane.pin_gt = pin;
        lane.est_voxel_on_main = upload && !on_device;
#if ME_TUNE_SUITE_SORT_FIRST
        if (upload && on_device && p->evaluate_mme) {
            if (!ctx->suite_event) ME_CHECK(ctx, hipEventCreateWithFlags(reinterpret_cast<hipEvent_t *>(&ctx->suite_event), hipEventDisableTiming));
            lane.sort_event = static_cast<hipEvent_t>(ctx->suite_event);
        }
#endif
        worker->post(&lane);
    }
    bool est_pinned = false, gt_pinned_here = false;
    // everything the main lane does; on failure the second lane is stopped and joined before returning
    auto main_lane = [&]() -> int {
        auto t0 = Clock::now();
        if (upload) {
            // ME_SUITE_PIN_HOST_INPUT: page-lock the caller's buffers for the duration of the call (pageable memory crosses PCIe
            // through the runtime's staging buffers at ~2/3 of the pinned rate); a buffer that is pinned already is left alone
            if (pin) est_pinned = hipHostRegister(const_cast<double *>(est), (size_t) n_est * 24, hipHostRegisterDefault) == hipSuccess;
            if (pin && !overlap) gt_pinned_here = hipHostRegister(const_cast<double *>(gt), (size_t) n_gt * 24, hipHostRegisterDefault) == hipSuccess;
            (void) hipGetLastError();  // (a refused registration is not an error of the call)
            ME_TRY(me::cloud_upload(ctx, ME_SLOT_EST, est, on_device, n_est, nullptr, p->nn_radius));
            if (overlap) lane.set(&SuiteLane::est_on_device);
            if (!overlap) ME_TRY(me::cloud_upload(ctx, ME_SLOT_GT, gt, on_device, n_gt, nullptr, p->nn_radius));
        } else if (overlap) {
            lane.set(&SuiteLane::est_on_device);
        }
        out->stage_ms[0] = ms_since(t0);
        if (overlap && !moved) lane.set(&SuiteLane::est_final);
        if (p->evaluate_mme) {
            double s = 0;
            long long nv = 0;
            t0 = Clock::now();
            if (overlap && lane.sort_event) {
                // (device-resident input only: the ground truth's sort is queued within a millisecond of the call's start — with host
                // input it follows a 20 ms copy, and the map's MME is what hides that copy)
                if (!lane.wait(&SuiteLane::sort_queued)) return lane.rc.load() != ME_OK ? lane.rc.load() : ME_ERR_STATE;
                ME_CHECK(ctx, hipStreamWaitEvent(ctx->stream, lane.sort_event, 0));
            }
            ME_TRY(me::mme_run(ctx, ME_SLOT_EST, p->nn_radius, 10, nullptr, nullptr, &s, &nv));  // k >= 10 (:1675)
            out->stage_ms[4] = ms_since(t0);
            out->mme_est = nv > 0 ? s / (double) nv : 0.0;
            out->mme_est_valid = nv;
        }
        if (moved) {  // *map_3d_ = map_3d_->Transform(initial_matrix) (:1206), after the MME of the map as loaded (:56)
            t0 = Clock::now();
            if (p->evaluate_mme) ME_TRY(me::mme_carry_out(ctx, ME_SLOT_EST));
            ME_TRY(me::cloud_transform(ctx, ME_SLOT_EST, T));
            if (p->evaluate_mme) ME_TRY(me::mme_carry_in(ctx, ME_SLOT_EST));
            out->stage_ms[0] += ms_since(t0);
            if (overlap) lane.set(&SuiteLane::est_final);
        }
        // the map's octree, left out of its index build: the MME above started that much earlier, and this lane would now wait for the
        // ground truth's cell tables anyway
        ME_TRY(me::cloud_finish_octree(ctx, ME_SLOT_EST));
        if (overlap) lane.set(&SuiteLane::est_tree);
        if (overlap && upload && !on_device) {
            // host input: the ground truth is still crossing PCIe (the map's index + MME are shorter than its copy) and this lane would
            // idle until it is indexed — the map's voxel table is built here instead of on the second lane after the ground truth's
            // (its voxel_build(est) then finds the table cached): that much less work is left when the last byte has arrived
            t0 = Clock::now();
            ME_TRY(me::voxel_build(ctx, ME_SLOT_EST, p->vmd_voxel_size, false));
            out->stage_ms[6] += ms_since(t0);
        }
        if (overlap) {
            lane.wait(&SuiteLane::gt_ready);
            if (lane.rc.load() != ME_OK) return lane.rc.load();
        }
        if (p->evaluate_mme && p->evaluate_gt_mme) {
            double s = 0;
            long long nv = 0;
            t0 = Clock::now();
            ME_TRY(me::mme_run(ctx, ME_SLOT_GT, p->nn_radius, 5, nullptr, nullptr, &s, &nv));  // k >= 5 (:1458)
            out->stage_ms[5] = ms_since(t0);
            out->mme_gt = nv > 0 ? s / (double) nv : 0.0;
            out->mme_gt_valid = nv;
        }
        // AC / COM both directions (:1213-1242) + full CD (:1398-1431) from the same two searches
        me_nn_partial pe{}, pg{};
        t0 = Clock::now();
        if (overlap) {
            lane.wait(&SuiteLane::gt_tree);
            if (lane.rc.load() != ME_OK) return lane.rc.load();
        } else {
            ME_TRY(me::cloud_finish_octree(ctx, ME_SLOT_GT));
        }
        ME_TRY(me::nn_search(ctx, ME_SLOT_EST, ME_SLOT_GT));
        ME_TRY(me::nn_partial(ctx, ME_SLOT_EST, p->icp_max_distance, p->gate_mode, p->trunc, &pe));
        out->stage_ms[1] = ms_since(t0);
        t0 = Clock::now();
#if ME_TUNE_SUITE_NN_FIRST
        if (overlap && !lane.est_voxel_on_main && !lane.est_voxel_taken.exchange(1)) {
            // (the second lane is still searching or on the ground truth's voxel table: the map's is built here instead of waiting)
            ME_TRY(me::voxel_build(ctx, ME_SLOT_EST, p->vmd_voxel_size, false));
            out->stage_ms[6] += ms_since(t0);
            t0 = Clock::now();
        }
#endif
        // the second pass of this lane's direction, and AWD / CDF / SCS (:85, :240-390) when the second lane has finished both voxel
        // tables already: neither needs the other direction's search, which the second lane is finishing meanwhile.  (Not earlier:
        // before this lane's search the chain of ~20 tiny launches waits behind the second lane's resident search waves — 3.7 ms —
        // and the two searches no longer overlap: 44.9 ms per step against 44.0, scratch/tl.sh.)
        double sig_e[5], sig_g[5];
        ME_TRY(sigma_pass(ctx, ME_SLOT_EST, p, pe, sig_e));
        out->stage_ms[3] = ms_since(t0);
        int64_t n_rows = 0;
        bool vmd_done = false;
        auto vmd = [&]() -> int {
            const auto tv = Clock::now();
            ME_TRY(me_awd_scs(ctx, p->vmd_voxel_size, p->min_pts > 0 ? p->min_pts : 100, p->scs_radius > 0 ? p->scs_radius : 5, nullptr,
                              nullptr, &n_rows, &out->awd, &out->scs, nullptr));
            out->stage_ms[6] += ms_since(tv);
            vmd_done = true;
            return ME_OK;
        };
        if (overlap && lane.is_set(&SuiteLane::tables_ready)) ME_TRY(vmd());
        t0 = Clock::now();
        if (overlap) {
            if (!lane.join()) return lane.rc.load();  // the second lane has searched the other direction meanwhile (and built both voxel tables)
            pg = lane.back;
            for (int k = 0; k < 5; ++k) sig_g[k] = lane.back_sig[k];
        } else {
            ME_TRY(me::nn_search(ctx, ME_SLOT_GT, ME_SLOT_EST));
            ME_TRY(me::nn_partial(ctx, ME_SLOT_GT, p->icp_max_distance, p->gate_mode, p->trunc, &pg));
            ME_TRY(sigma_pass(ctx, ME_SLOT_GT, p, pg, sig_g));
        }
        out->stage_ms[2] = ms_since(t0);
        me_nn_finalize(&pe, sig_e, ctx->cloud[ME_SLOT_EST].n, &out->est_gt);
        me_nn_finalize(&pg, sig_g, ctx->cloud[ME_SLOT_GT].n, &out->gt_est);
        out->full_chamfer = out->est_gt.mean_nn_dist + out->gt_est.mean_nn_dist;  // computeChamferDistance (:1429)
        // (the voxel tables are cached on the clouds when the second lane built them)
        if (!vmd_done) ME_TRY(vmd());
        out->n_w_voxels = n_rows;
        return ME_OK;
    };
    int rc;
    try {  // (a std::bad_alloc from a host-side vector must not cross the C boundary — and not leave the lane running)
        rc = main_lane();
    } catch (const std::bad_alloc &) {
        rc = ctx->fail(ME_ERR_HIP, "me_run_suite_from: out of host memory");
    } catch (const std::exception &e) {
        rc = ctx->fail(ME_ERR_HIP, std::string("me_run_suite_from: ") + e.what());
    }
    if (overlap) {
        lane.abort_and_join();
        if (rc != ME_OK && rc == lane.rc.load() && lane.t) ctx->err = lane.t->err;  // the second lane's failure: its message
    }
    if (est_pinned) (void) hipHostUnregister(const_cast<double *>(est));
    if (gt_pinned_here || lane.gt_pinned) (void) hipHostUnregister(const_cast<double *>(gt));
    out->stage_ms[7] = ms_since(t_all);
    return rc;
}

}  // extern "C"
