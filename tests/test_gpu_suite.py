"""me_run_suite_from — the whole pass of MapEval::process() (map_eval.cpp:52-85: computeMME :56, calculateMetricsWithInitialMatrix
:76, calculateVMD :85) in ONE C-ABI call with the library's internal second lane (csrc/me_suite.hip).

What is checked: the overlapped call == the sequential call == the separate calls, bit for bit; against the oracle (the CPU
restatement of the reference) within the suite's tolerances; the reference's order for a non-identity initial_matrix (MME on the map
as loaded, :56, the transform afterwards, :1206); the per-point products it leaves on the device (me_mme_fetch, me_nn_fetch); host
and device input; error paths (a failing second lane must not hang the caller)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from cloud_map_evaluation_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def pair():
    from cloud_map_evaluation_amd import synth

    est, gt = synth.multisession_pair(400_000, 3, density=2500.0, seed=11)
    return est.numpy(), gt.numpy()


def _param(T=None):
    from cloud_map_evaluation_amd.engine import Param

    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=3.0)
    if T is not None:
        P.initial_matrix_ = np.asarray(T, dtype=np.float64)
    return P


def _same(a, b):
    for k in ("full_chamfer", "mme_est", "mme_gt", "mme_est_valid", "mme_gt_valid", "awd", "scs", "n_w_voxels"):
        assert getattr(a, k) == getattr(b, k), k
    for side in ("est_gt", "gt_est"):
        x, y = getattr(a, side), getattr(b, side)
        assert x.n_src == y.n_src and x.n_corr == y.n_corr and x.mean_nn_dist == y.mean_nn_dist
        for k in ("mean", "rmse", "fitness", "sigma", "number"):
            assert list(getattr(x, k)) == list(getattr(y, k)), (side, k)


def test_overlapped_call_equals_sequential_call_equals_the_separate_calls(eng, pair):
    est, gt = pair
    P = _param()
    # the separate calls, as the round-1 hosts make them
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    ref = eng.run_suite(P)
    ent_ref = eng.mme(0, 0.1, 10, per_point=True)
    seq = eng.run_suite_from(est, gt, P, overlap=False)
    _same(ref, seq)
    for _ in range(3):  # (a schedule bug would show as a flaky difference)
        ovl = eng.run_suite_from(est, gt, P, overlap=True)
        _same(ref, ovl)
    assert ovl.stage_ms[7] > 0 and ovl.stage_ms[0] > 0
    # per-point products left on the device by the one call
    ent, val = eng.mme_fetch(0)
    assert np.array_equal(ent, ent_ref[1]) and np.array_equal(val, ent_ref[2])
    idx, d2 = eng.nn_fetch(0)
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    idx2, d22 = eng.nn1(0, 1)
    assert np.array_equal(idx, idx2) and np.array_equal(d2, d22)


def test_device_input_and_resident_clouds(eng, pair):
    import torch

    est, gt = pair
    P = _param()
    host = eng.run_suite_from(est, gt, P, overlap=True)
    pinned = eng.run_suite_from(est, gt, P, overlap=True, pin_host_input=True)  # ME_SUITE_PIN_HOST_INPUT: registered for the call
    _same(host, pinned)
    pinned = eng.run_suite_from(est, gt, P, overlap=False, pin_host_input=True)
    _same(host, pinned)
    assert np.isfinite(est).all()  # (still readable: unregistered again)
    dev = torch.device("cuda", 0)
    est_d, gt_d = torch.from_numpy(est).to(dev), torch.from_numpy(gt).to(dev)
    a = eng.run_suite_from(est_d, gt_d, P, overlap=True)
    _same(host, a)
    assert torch.equal(est_d.cpu(), torch.from_numpy(est))  # the caller's buffers are left alone
    b = eng.run_suite_from(None, None, P, overlap=True)      # the clouds already uploaded
    _same(host, b)
    c = eng.run_suite_from(None, None, P, overlap=False)
    _same(host, c)


def test_against_the_oracle(eng, pair):
    import oracle

    est, gt = pair
    P = _param()
    o = eng.run_suite_from(est, gt, P, overlap=True)
    eg = oracle.reg_stats(est, gt, P.icp_max_distance_, 0, P.trunc_dist_, threads=0)
    ge = oracle.reg_stats(gt, est, P.icp_max_distance_, 0, P.trunc_dist_, threads=0)
    assert o.est_gt.n_corr == eg.n_corr and o.gt_est.n_corr == ge.n_corr
    assert [int(x) for x in o.est_gt.number] == [int(x) for x in eg.number]  # inlier counts: bit-exact
    assert [int(x) for x in o.gt_est.number] == [int(x) for x in ge.number]
    np.testing.assert_allclose(list(o.est_gt.rmse), eg.rmse, rtol=1e-9)
    np.testing.assert_allclose(list(o.gt_est.sigma), ge.sigma, rtol=1e-9)
    cd = eg.sum_sqrt_all / len(est) + ge.sum_sqrt_all / len(gt)
    assert abs(o.full_chamfer - cd) <= 1e-9 * cd
    m_e = oracle.mme(est, P.nn_radius_, 10, mode=2)
    m_g = oracle.mme(gt, P.nn_radius_, 5, mode=0)
    assert o.mme_est_valid == m_e[3] and o.mme_gt_valid == m_g[3]
    assert abs(o.mme_est - m_e[0]) <= 1e-9 * abs(m_e[0]) and abs(o.mme_gt - m_g[0]) <= 1e-9 * abs(m_g[0])
    v = oracle.awd_scs(oracle.VoxelMap(gt, P.vmd_voxel_size_), oracle.VoxelMap(est, P.vmd_voxel_size_))
    assert o.n_w_voxels == len(v["w_sorted"])
    assert abs(o.awd - v["awd"]) <= 1e-9 * abs(v["awd"]) and abs(o.scs - v["scs"]) <= 1e-9 * abs(v["scs"])


def test_initial_matrix_follows_the_reference_order(eng, pair):
    """MME on the map AS LOADED (map_eval.cpp:56), *map_3d_ = Transform(initial_matrix) afterwards (:1206)."""
    est, gt = pair
    c, s = np.cos(0.01), np.sin(0.01)
    T = np.array([[c, -s, 0, 0.03], [s, c, 0, -0.02], [0, 0, 1, 0.01], [0, 0, 0, 1.0]])
    P = _param(T)
    one = eng.run_suite_from(est, gt, P, overlap=True)
    seq = eng.run_suite_from(est, gt, P, overlap=False)
    _same(one, seq)
    moved = eng.download(0)  # the map as it lies on the device after the call: transformed
    # the separate calls in the reference's order
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    mme_est = eng.mme(0, 0.1, 10, per_point=False)
    eng.transform_cloud(0, T)
    assert np.array_equal(moved, eng.download(0))
    ref = eng.run_suite(_param())
    assert one.mme_est == mme_est[0] and one.mme_est_valid == mme_est[3]      # MME of the untransformed map
    assert one.mme_gt == ref.mme_gt and one.full_chamfer == ref.full_chamfer and one.awd == ref.awd and one.scs == ref.scs
    assert list(one.est_gt.rmse) == list(ref.est_gt.rmse) and list(one.gt_est.number) == list(ref.gt_est.number)


def test_switches_and_error_paths(eng, pair):
    from cloud_map_evaluation_amd.engine import MapEvalError

    est, gt = pair
    P = _param()
    P.evaluate_gt_mme_ = False
    o = eng.run_suite_from(est, gt, P, overlap=True)
    assert o.mme_gt == 0.0 and o.mme_gt_valid == 0 and o.mme_est_valid > 0
    P.evaluate_mme_ = False
    o = eng.run_suite_from(est, gt, P, overlap=True)
    assert o.mme_est == 0.0 and o.mme_est_valid == 0 and o.awd > 0
    # a ground truth with a non-finite coordinate fails on the SECOND lane: the call returns its error (it must not hang or crash)
    bad = gt.copy()
    bad[123, 1] = -np.inf
    with pytest.raises(MapEvalError, match="NaN"):
        eng.run_suite_from(est, bad, _param(), overlap=True)
    # ... and a bad map on the main lane, with the second lane already running
    bad = est.copy()
    bad[5, 0] = np.inf
    with pytest.raises(MapEvalError, match="NaN"):
        eng.run_suite_from(bad, gt, _param(), overlap=True)
    with pytest.raises(MapEvalError):
        eng.run_suite_from(est[:0], gt, _param(), overlap=True)  # empty map (map_eval.cpp:32-35 returns -1)
    # the engine is still usable
    ok = eng.run_suite_from(est, gt, _param(), overlap=True)
    assert ok.mme_est_valid > 0


def test_bench_size_step_through_the_one_call_matches_the_python_driver():
    """5 M + 5 M, device-resident, borrowed input: the C call and the Python-driven two-lane step give the same scalars."""
    import torch

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import Engine

    dev = torch.device("cuda", 0)
    est, gt = synth.multisession_pair(5_000_000, 3, density=2500.0, seed=100, device=dev)
    P = _param()
    with Engine(0, borrow_device_input=True) as e:
        a = medist.suite_step(e, None, dev, est, gt, P, True, overlap=True)
        b = Engine.suite_dict(e.run_suite_from(est, gt, P, overlap=True))
    for k in ("cd", "mme_est", "mme_gt", "mme_valid", "awd", "scs", "n_w", "n_est", "n_gt"):
        assert a[k] == b[k], k
    for k in ("rmse", "fitness", "sigma", "mean", "number"):
        assert np.array_equal(a["est_gt"][k], b["est_gt"][k]) and np.array_equal(a["gt_est"][k], b["gt_est"][k])


def test_an_index_is_complete_on_the_device_when_its_upload_returns():
    """Round 5 regression (profiles/EXPERIMENTS.md "A cross-stream race"): me_upload_cloud used to return with the cell tables and the
    octree still in flight on its stream; a second lane (me_twin) that started a kernel on ITS stream right away probed a half-built
    hash table — silent with a persistent engine (the stale bytes were the previous, identical table), minutes of spinning in a
    fresh context whose buffers hold recycled bytes.  Fresh engines, fresh host arrays, the hand-over as tight as Python allows:
    the other lane's results must be the oracle's every time, and no iteration may take seconds."""
    import time

    import oracle
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import Engine

    est, gt = synth.multisession_pair(1_500_000, 3, density=2500.0, seed=23)
    est, gt = est.numpy(), gt.numpy()
    o = oracle.mme(gt, 0.1, 5, mode=0)
    for it in range(3):
        a, b = est.copy(), gt.copy()
        with Engine(0) as e:
            lane = e.twin()
            e.upload(0, a, cell_size=0.1)      # (dirties the allocator's recycled blocks for the next iteration)
            t0 = time.perf_counter()
            lane.upload(1, b, cell_size=0.1)   # the index is built on the twin's stream ...
            m = e.mme(1, 0.1, 5)               # ... and read by a kernel on the primary stream at once
            idx, d2 = e.nn1(0, 1)
            dt = time.perf_counter() - t0
        assert m[3] == o[3] and np.array_equal(m[2], o[2]), it
        assert dt < 5.0, f"iteration {it} took {dt:.1f} s"
        del a, b


def test_resident_clouds_with_an_unfit_index_are_rebuilt_before_the_lanes_start(pair):
    """clouds uploaded with a cell size that does not fit the radius (0.5 m here; the "automatic" extent / 128 of
    cell_size <= 0 is another such value for all but the smallest scenes — for THIS pair's ground truth it happens to be 0.108 m, which
    fits r = 0.1 and is therefore kept: another lattice, other wavefronts, entropies equal to ~1e-11 instead of bit for bit) and then
    evaluated with est = gt = NULL.  mme_run rebuilds such an index — re-sorting `sp`, reallocating the cell tables and the octree — and with
    ME_SUITE_OVERLAP the second lane used to read the same Cloud at that moment (its voxel_build saw index_valid == false and started a
    second rebuild).  The indexes are now settled on one lane before the second one starts: the overlapped call on resident clouds
    equals the sequential one and the call from the raw clouds, every time, with and without the ground truth's MME."""
    from cloud_map_evaluation_amd.engine import Engine

    est, gt = pair
    for gt_mme in (True, False):
        P = _param()
        P.evaluate_gt_mme_ = gt_mme
        with Engine(0) as e:
            want = e.run_suite_from(est, gt, P, overlap=False)
            for it in range(3):
                e.upload(0, est, cell_size=0.5)
                e.upload(1, gt, cell_size=0.5)
                got = e.run_suite_from(None, None, P, overlap=True)
                _same(want, got)
            e.upload(0, est, cell_size=0.5)
            e.upload(1, gt, cell_size=0.5)
            _same(want, e.run_suite_from(None, None, P, overlap=False))
            # "automatic" cells: rebuilt where they do not fit, kept where they do — every count equal, sums to rounding
            e.upload(0, est, cell_size=0.0)
            e.upload(1, gt, cell_size=0.0)
            auto = e.run_suite_from(None, None, P, overlap=True)
            assert (auto.mme_est_valid, auto.mme_gt_valid, auto.n_w_voxels) == (want.mme_est_valid, want.mme_gt_valid, want.n_w_voxels)
            assert list(auto.est_gt.number) == list(want.est_gt.number) and list(auto.gt_est.number) == list(want.gt_est.number)
            np.testing.assert_allclose([auto.mme_est, auto.mme_gt, auto.full_chamfer, auto.awd, auto.scs],
                                       [want.mme_est, want.mme_gt, want.full_chamfer, want.awd, want.scs], rtol=1e-11)
    # no MME at all: the stages need SOME index only (built by the uploads above; an invalidated one is rebuilt up front as well)
    P = _param()
    P.evaluate_mme_ = False
    with Engine(0) as e:
        want = e.run_suite_from(est, gt, P, overlap=False)
        e.upload(0, est, cell_size=0.5)
        e.upload(1, gt, cell_size=0.5)
        _same(want, e.run_suite_from(None, None, P, overlap=True))


def test_pinned_torch_buffers_with_pin_host_input_on_both_lanes(eng, pair):
    """buffers that are page-locked ALREADY (torch pinned tensors) with ME_SUITE_PIN_HOST_INPUT | ME_SUITE_OVERLAP —
    hipHostRegister refuses them ("already pinned: no-op" in the header) on the second lane's own thread, and the refusal used to stay
    pending there (hipGetLastError is sticky per thread on ROCm 7) until the lane's next ME_CHECK(hipGetLastError()) turned it into
    ME_ERR_HIP for the whole call."""
    import torch

    est, gt = pair
    P = _param()
    want = eng.run_suite_from(est, gt, P, overlap=True)
    est_p, gt_p = torch.from_numpy(est).pin_memory(), torch.from_numpy(gt).pin_memory()
    for overlap in (True, False, True):
        got = eng.run_suite_from(est_p, gt_p, P, overlap=overlap, pin_host_input=True)
        _same(want, got)
    assert torch.equal(est_p, torch.from_numpy(est)) and est_p.is_pinned()  # (still pinned: the call did not unregister what it did not register)


def test_the_lane_thread_belongs_to_the_context(pair):
    """Round 6: the second lane's host thread is created by the first overlapped call of a context, reused by the later ones and joined by
    me_destroy — many calls on one engine, then engines created and destroyed in a loop (with and without an overlapped call), must
    neither leak threads nor hang."""
    import threading

    from cloud_map_evaluation_amd.engine import Engine

    est, gt = pair
    est, gt = est[:100_000], gt[:100_000]
    P = _param()
    with Engine(0) as e:
        want = e.run_suite_from(est, gt, P, overlap=False)
        for _ in range(10):
            _same(want, e.run_suite_from(est, gt, P, overlap=True))
    n0 = threading.active_count()
    for i in range(6):
        with Engine(0) as e:
            if i % 2 == 0:
                _same(want, e.run_suite_from(est, gt, P, overlap=True))
    assert threading.active_count() == n0
    import os

    native = len(os.listdir("/proc/self/task"))
    for i in range(6):
        with Engine(0) as e:
            _same(want, e.run_suite_from(est, gt, P, overlap=True))
    assert len(os.listdir("/proc/self/task")) <= native + 1  # (native threads: every context's worker was joined)
