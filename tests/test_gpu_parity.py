"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact counts (inliers, valid points, voxel populations, correspondences) and
bit-exact squared NN distances; floating-point metrics within 1e-5 relative (we assert far tighter, RTOL below).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9  # fp metrics: observed ~1e-13; the contract is 1e-5
TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from cloud_map_evaluation_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def cube():
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(100_000, seed=42)  # BASELINE config 0 (C1)
    return est.numpy(), gt.numpy()


@pytest.fixture(scope="module")
def campus():
    from cloud_map_evaluation_amd import synth

    est, gt = synth.campus_pair(300_000, density=2500.0, seed=100, origin=(812.5, -433.25, 37.0))
    return est.numpy(), gt.numpy()


def _stats_equal(got, exp):
    assert got.n_src == exp.n_src and got.n_corr == exp.n_corr
    assert np.array_equal(got.number, exp.number)  # bit-exact inlier counts
    assert np.array_equal(got.fitness, exp.fitness)
    for k in ("mean", "rmse", "sigma"):
        np.testing.assert_allclose(getattr(got, k), getattr(exp, k), rtol=RTOL)


# ---------------------------------------------------------------------------------------------------------
def test_nn1_bit_exact_both_directions(eng, cube):
    import oracle

    est, gt = cube
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    for q, r, qa, ra in ((0, 1, est, gt), (1, 0, gt, est)):
        idx, d2 = eng.nn1(q, r)
        oi, od2 = oracle.nn1(ra, qa)
        assert np.array_equal(d2, od2)  # bit-exact squared distances
        assert np.array_equal(idx, oi)  # same neighbour (ties -> smallest index in both)


def test_nn1_large_offsets_far_queries_and_duplicates(eng, campus):
    import oracle

    est, gt = campus
    rng = np.random.default_rng(3)
    far = rng.uniform(-500, 500, (2000, 3)) + gt.mean(0)  # queries far outside the reference cloud
    q = np.concatenate([est[:50_000], far, gt[:1000], gt[:1000]])  # exact duplicates of ref points too
    eng.upload(0, q)
    eng.upload(1, np.concatenate([gt, gt[:5000]]))  # duplicated reference points
    idx, d2 = eng.nn1(0, 1)
    oi, od2 = oracle.nn1(np.concatenate([gt, gt[:5000]]), q)
    assert np.array_equal(d2, od2)
    assert np.array_equal(idx, oi)
    assert np.all(d2[-2000:] == 0.0)


@pytest.mark.parametrize("n_ref", [1, 2, 15, 16, 17, 127, 129, 1025, 2047, 2048, 2049, 4097])  # (2048: the block of the cell-table passes)
def test_nn1_tiny_and_ragged_reference_sizes(eng, n_ref):
    import oracle

    rng = np.random.default_rng(n_ref)
    ref = rng.normal(0, 3, (n_ref, 3))
    q = rng.normal(0, 4, (333, 3))
    eng.upload(0, q)
    eng.upload(1, ref)
    idx, d2 = eng.nn1(0, 1)
    oi, od2 = oracle.nn1(ref, q)
    assert np.array_equal(d2, od2) and np.array_equal(idx, oi)


@pytest.mark.parametrize("gate,mode", [(1.0, 0), (0.0025, 0), (0.05, 1), (-1.0, 0)])
def test_ac_com_stats_parity(eng, cube, gate, mode):
    import oracle

    est, gt = cube
    eng.upload(0, est)
    eng.upload(1, gt)
    eng.nn1(0, 1, fetch=False)
    _stats_equal(eng.nn_stats(0, gate, mode, TRUNC), oracle.reg_stats(est, gt, gate, mode, TRUNC))
    eng.nn1(1, 0, fetch=False)
    _stats_equal(eng.nn_stats(1, gate, mode, TRUNC), oracle.reg_stats(gt, est, gate, mode, TRUNC))


def test_initial_matrix_transform_and_metrics(eng, cube):
    import oracle
    from cloud_map_evaluation_amd.engine import Param

    est, gt = cube
    th = 0.01
    T = np.eye(4)
    T[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    T[:3, 3] = [0.01, -0.02, 0.005]
    eng.upload(0, est, T=T)
    eng.upload(1, gt)
    est_t = oracle.transform(est, T)
    assert np.array_equal(eng.download(0), est_t)  # bit-exact transform (map_eval.cpp:1206)
    p = Param(icp_max_distance_=1.0, trunc_dist_=TRUNC)
    eg, ge, cd_vec = eng.calculateMetricsWithInitialMatrix(p)
    oeg = oracle.reg_stats(est_t, gt, 1.0, 0, TRUNC)
    oge = oracle.reg_stats(gt, est_t, 1.0, 0, TRUNC)
    _stats_equal(eg, oeg)
    _stats_equal(ge, oge)
    np.testing.assert_allclose(cd_vec, oeg.rmse + oge.rmse, rtol=RTOL)


def test_chamfer_parity(eng, campus):
    import oracle

    est, gt = campus
    eng.upload(0, est)
    eng.upload(1, gt)
    np.testing.assert_allclose(eng.computeChamferDistance(), oracle.chamfer(est, gt), rtol=RTOL)


def test_empty_correspondence_set_is_nan(eng):
    a = np.random.default_rng(0).uniform(0, 1, (100, 3))
    eng.upload(0, a)
    eng.upload(1, a + 50.0)
    eng.nn1(0, 1, fetch=False)
    s = eng.nn_stats(0, 1.0, 0, TRUNC)
    assert s.n_corr == 0 and np.all(np.isnan(s.mean)) and np.all(np.isnan(s.rmse)) and np.all(s.fitness == 0)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("min_k", [10, 5])
def test_mme_parity_cube(eng, cube, min_k):
    import oracle

    est, gt = cube
    cloud = est if min_k == 10 else gt
    eng.upload(0, cloud, cell_size=0.1)
    mean, ent, val, nv, s = eng.mme(0, 0.1, min_k)
    omean, oent, oval, onv, osum = oracle.mme(cloud, 0.1, min_k)
    assert nv == onv and np.array_equal(val, oval)  # bit-exact validity
    assert onv > 0.5 * len(cloud)
    np.testing.assert_allclose(ent, oent, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mean, omean, rtol=RTOL)
    np.testing.assert_allclose(s, osum, rtol=RTOL)


def test_mme_parity_campus_offset_and_radius_change(eng, campus):
    import oracle

    est, _ = campus
    sub = est[:150_000]
    eng.upload(0, sub)  # automatic cell size -> me_mme rebuilds the grid for the radius
    for r, min_k in ((0.1, 10), (0.17, 10), (0.05, 5)):
        mean, ent, val, nv, s = eng.mme(0, r, min_k)
        omean, oent, oval, onv, osum = oracle.mme(sub, r, min_k)
        assert nv == onv and np.array_equal(val, oval)
        np.testing.assert_allclose(ent, oent, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(mean, omean, rtol=RTOL)


def test_mme_no_valid_points_returns_zero(eng):
    rng = np.random.default_rng(0)
    eng.upload(0, rng.uniform(0, 100, (500, 3)))
    mean, ent, val, nv, s = eng.mme(0, 0.1, 10)
    assert mean == 0.0 and nv == 0 and not val.any() and not ent.any()


# ---------------------------------------------------------------------------------------------------------
def test_voxel_gaussians_parity(eng, campus):
    import oracle

    est, gt = campus
    eng.upload(1, gt)
    keys, n, mu, sig, ent = eng.voxel_gaussians(1, 3.0)
    okeys, on, omu, osig, oent = oracle.VoxelMap(gt, 3.0).export()
    assert np.array_equal(keys, okeys) and np.array_equal(n, on)  # bit-exact voxel populations
    np.testing.assert_allclose(mu, omu, rtol=1e-13)
    from tests._tol import assert_sigma_close

    assert_sigma_close(sig, osig)  # 1e-9 of each matrix's scale (two-pass vs the reference's streaming Welford)
    # every voxel the reference evaluates: its post-pass computes the entropy for n > 10 only (voxel_calculator.cpp:46-50), the
    # others keep the constructor's 0 — compared too.  0.5 ln((2 pi e)^3 det) is compared absolutely: a relative
    # 1e-9 on Sigma (two-pass vs streaming Welford) is an absolute ~1e-9 on the logarithm whatever its size
    big = on > 10
    assert big.any()
    np.testing.assert_allclose(ent[big], oent[big], rtol=0, atol=1e-8)
    assert np.array_equal(ent[~big], oent[~big])


def test_voxel_gaussians_of_scattered_points_take_the_three_pass_build(eng):
    """Round 6: the one-pass voxel build gives a row of 64 sorted points two record slots and room for about as many further runs in
    its overflow regions.  A cloud whose curve-neighbours all lie in different voxels — uniform points in a large box, a 1 m grid —
    overflows them (~64 runs per row): the library must notice and take the three-pass build; keys, populations, means and covariances
    against the oracle as for any other cloud, with mixed populations (a dense blob on top)."""
    import oracle

    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-60.0, 60.0, (150_000, 3)), rng.normal(0.0, 0.8, (60_000, 3)) + np.array([7.3, -2.1, 4.4])])
    eng.upload(1, pts)
    keys, n, mu, sig, ent = eng.voxel_gaussians(1, 1.0)
    okeys, on, omu, osig, oent = oracle.VoxelMap(pts, 1.0).export()
    assert np.array_equal(keys, okeys) and np.array_equal(n, on) and n.max() > 100 and (n == 1).sum() > 10_000
    np.testing.assert_allclose(mu, omu, rtol=1e-13, atol=1e-13)
    from tests._tol import assert_sigma_close

    assert_sigma_close(sig, osig)
    big = on > 10
    np.testing.assert_allclose(ent[big], oent[big], rtol=0, atol=1e-8)


@pytest.mark.parametrize("vs", [3.0, 0.5])
def test_awd_cdf_scs_parity(eng, campus, cube, vs):
    import oracle

    est, gt = campus if vs == 3.0 else cube
    eng.upload(0, est)
    eng.upload(1, gt)
    res = eng.calculateVMD(vs)
    ores = oracle.awd_scs(oracle.VoxelMap(gt, vs), oracle.VoxelMap(est, vs))
    assert res["counts"] == ores["counts"]
    assert res["rows"].shape == ores["rows"].shape and len(res["rows"]) > 20
    assert np.array_equal(res["rows"][:, :6], ores["rows"][:, :6])        # same voxels, same order
    assert np.array_equal(res["rows"][:, 10:12], ores["rows"][:, 10:12])  # n_gt, n_est
    from tests._tol import assert_voxel_rows_close

    assert_voxel_rows_close(res["rows"], ores["rows"], rtol=1e-8)
    np.testing.assert_allclose(res["rows"][:, 9], ores["rows"][:, 9], rtol=1e-8)  # W per voxel
    np.testing.assert_allclose(res["w_sorted"], ores["w_sorted"], rtol=1e-8)
    np.testing.assert_allclose(res["awd"], ores["awd"], rtol=RTOL)
    np.testing.assert_allclose(res["scs"], ores["scs"], rtol=RTOL)


def test_awd_empty_is_nan(eng):
    a = np.random.default_rng(0).uniform(0, 1, (500, 3))
    eng.upload(0, a + 100)
    eng.upload(1, a)
    res = eng.calculateVMD(0.5)
    assert np.isnan(res["awd"]) and np.isnan(res["scs"]) and res["n_rows"] == 0


def test_golden_reference_run_through_device_kernels(eng, golden):
    """The reference's own run output replayed through the DEVICE W2 and SCS kernels."""
    rows = golden["rows"]

    def full(r6):
        m = np.empty((len(r6), 9))
        m[:, 0], m[:, 1], m[:, 2] = r6[:, 0], r6[:, 1], r6[:, 2]
        m[:, 3], m[:, 4], m[:, 5] = r6[:, 1], r6[:, 3], r6[:, 4]
        m[:, 6], m[:, 7], m[:, 8] = r6[:, 2], r6[:, 4], r6[:, 5]
        return m

    w = eng.w2_batch(rows[:, 18:21], full(rows[:, 21:27]), rows[:, 10].astype(np.int32),
                     rows[:, 6:9], full(rows[:, 12:18]), rows[:, 11].astype(np.int32))
    rel = np.abs(w - rows[:, 9]) / np.maximum(rows[:, 9], 1e-12)
    assert np.median(rel) < 2e-3 and np.quantile(rel, 0.99) < 5e-2  # print precision of the fixture (6 digits)
    import oracle

    ow = np.array([oracle.w2_gaussian(rows[i, 18:21], full(rows[i:i + 1, 21:27])[0], int(rows[i, 10]), rows[i, 6:9],
                                      full(rows[i:i + 1, 12:18])[0], int(rows[i, 11])) for i in range(0, len(rows), 7)])
    np.testing.assert_allclose(w[::7], ow, rtol=1e-9, atol=1e-12)  # device == oracle on the golden inputs
    keys = np.rint(rows[:, 0:3] / float(golden["voxel_size"])).astype(np.int32)
    scs = eng.scs_table(keys, rows[:, 9], 5)
    assert abs(scs - float(golden["screenshot_scs"])) < 5e-6       # README screenshot SCS: 0.78121 (from the file's own W column)
    # ... and the DEVICE's numbers against the screenshot: AWD = the mean of the W the device kernel computed
    # from the file's mu / Sigma columns (6 printed digits: per-voxel W moves by ~1e-3 relative, the mean by 9e-7), SCS from those W
    assert abs(float(w.mean()) - float(golden["screenshot_vmd"])) < 5e-6   # screenshot VMD: 0.35303
    scs_dev = eng.scs_table(keys, w, 5)
    assert abs(scs_dev - float(golden["screenshot_scs"])) < 3e-5          # (0.781196 from the reprinted inputs)
    assert abs(rows[:, 9].mean() - float(golden["screenshot_vmd"])) < 5e-6  # (the fixture itself is the screenshot's run)


# ---------------------------------------------------------------------------------------------------------
def test_run_suite_matches_piecewise_and_oracle(eng, cube):
    import oracle
    from cloud_map_evaluation_amd.engine import Param

    est, gt = cube
    p = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=0.5)
    eng.upload(0, est, cell_size=p.nn_radius_)
    eng.upload(1, gt, cell_size=p.nn_radius_)
    out = eng.run_suite(p)
    np.testing.assert_allclose(out.full_chamfer, oracle.chamfer(est, gt), rtol=RTOL)
    np.testing.assert_allclose(out.mme_est, oracle.mme(est, 0.1, 10)[0], rtol=RTOL)
    np.testing.assert_allclose(out.mme_gt, oracle.mme(gt, 0.1, 5)[0], rtol=RTOL)
    o = oracle.reg_stats(est, gt, 1.0, 0, TRUNC)
    assert [out.est_gt.number[k] for k in range(5)] == list(o.number)
    np.testing.assert_allclose([out.est_gt.rmse[k] for k in range(5)], o.rmse, rtol=RTOL)
    ores = oracle.awd_scs(oracle.VoxelMap(gt, 0.5), oracle.VoxelMap(est, 0.5))
    np.testing.assert_allclose(out.awd, ores["awd"], rtol=RTOL)
    np.testing.assert_allclose(out.scs, ores["scs"], rtol=RTOL)
    assert out.n_w_voxels == len(ores["rows"])


def test_sharded_partials_sum_to_the_whole(eng, cube):
    """me_set_shard: the slabs of a 4-way split, run one after another on this GPU, add up to the 1-GPU result."""
    est, gt = cube
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    eng.nn1(0, 1, fetch=False)
    whole = eng.nn_stats(0, 1.0, 0, TRUNC)
    wm = eng.mme(0, 0.1, 10)
    tot_n = np.zeros(6, np.int64)
    tot_d = np.zeros(11)
    mme_s, mme_c = 0.0, 0
    ent = np.zeros(len(est))
    try:
        for r in range(4):
            eng.set_shard(r, 4)
            eng.nn1(0, 1, fetch=False)
            pp = eng.nn_partial_sums(0, 1.0, 0, TRUNC)
            tot_n += [pp.n_corr] + list(pp.n_inl)
            tot_d += list(pp.sum_d) + list(pp.sum_d2) + [pp.sum_sqrt_all]
            m = eng.mme(0, 0.1, 10)
            mme_s += m[4]
            mme_c += m[3]
            ent += m[1]
    finally:
        eng.set_shard(0, 1)
    assert tot_n[0] == whole.n_corr and list(tot_n[1:]) == list(whole.number.astype(np.int64))
    np.testing.assert_allclose(tot_d[:5] / whole.n_corr, whole.mean, rtol=1e-12)
    assert mme_c == wm[3]
    np.testing.assert_allclose(mme_s, wm[4], rtol=1e-12)
    # every point's entropy is computed by exactly one shard.  Since round 5 the moments are accumulated about the point of the
    # wavefront's group leader (me_mme.hip): a shard boundary shifts which 64 points share a wavefront, so the same entropy comes out
    # of differently rounded sums — equal to ~1e-13, the flags identical
    assert np.array_equal(ent != 0.0, wm[1] != 0.0)
    np.testing.assert_allclose(ent, wm[1], rtol=0, atol=1e-11)


def test_device_sqrt_is_correctly_rounded(eng):
    """Sum of sqrt(d2) feeds CD; spot-check the device sqrt against numpy on the fetched distances."""
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 10, (20000, 3))
    b = rng.uniform(0, 10, (20000, 3))
    eng.upload(0, a)
    eng.upload(1, b)
    _, d2 = eng.nn1(0, 1)
    p = eng.nn_partial_sums(0, -1.0, 0, TRUNC)
    np.testing.assert_allclose(p.sum_sqrt_all, np.sqrt(d2).sum(), rtol=1e-13)


# ---------------------------------------------------------------------------------------------------------
# "next" row of SURVEY.md section 8f, rank 1: VoxelDownSample (map_eval.cpp:38-39) and the in-place transform (:1206)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("vs", [0.05, 0.01, 0.37])
def test_voxel_downsample_bit_exact(eng, campus, vs):
    import oracle

    est, _ = campus
    eng.upload(0, est)
    n = eng.voxel_downsample(0, vs)
    got = eng.download(0)
    exp = oracle.voxel_downsample(est, vs)
    assert n == len(exp) == len(got)
    assert np.array_equal(got, exp)  # same voxels, same order, means accumulated in cloud order: bit-identical
    # the down-sampled cloud is a fully indexed cloud: metrics run on it
    eng.upload(1, exp)
    _, d2 = eng.nn1(0, 1)
    assert np.all(d2 == 0.0)


def test_voxel_downsample_too_small_voxel_is_an_error(eng):
    from cloud_map_evaluation_amd.engine import MapEvalError

    eng.upload(0, np.array([[0.0, 0, 0], [5000.0, 1, 1], [1.0, 2, 3]]))
    with pytest.raises(MapEvalError):
        eng.voxel_downsample(0, 1e-4)  # > 2^21 voxels per axis (Open3D: "voxel_size is too small")


def test_transform_cloud_in_place(eng, cube):
    import oracle

    est, gt = cube
    T = np.eye(4)
    T[:3, :3] = [[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]
    T[:3, 3] = [0.5, -0.25, 0.125]
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    before = eng.mme(0, 0.1, 10)
    eng.transform_cloud(0, T)
    est_t = oracle.transform(est, T)
    assert np.array_equal(eng.download(0), est_t)
    after = eng.mme(0, 0.1, 10)  # MME is invariant under a rigid transform (up to rounding)
    assert after[3] == before[3]
    np.testing.assert_allclose(after[0], before[0], rtol=1e-9)
    _, d2 = eng.nn1(0, 1)
    assert np.array_equal(d2, oracle.nn1(gt, est_t)[1])


def test_nn1_cascade_over_several_grid_levels_is_bit_exact():
    """Round 4: on a cloud much denser than the search cell the fine 1-NN grid sits several levels below the radius grid and
    the queries it cannot settle go through one list pass per coarser level before the octree (me_nn.hip, k_nn_grid<FROM_LIST>).
    A 3 m patch at ~10^5 pts/m^2 with drifted, noisy, thinned queries and outliers: the passes must have run, and every
    squared distance and neighbour index must be the oracle's."""
    import oracle

    from cloud_map_evaluation_amd.engine import Engine

    rng = np.random.default_rng(41)
    n = 900_000
    xy = rng.uniform(0.0, 3.0, (n, 2))
    gt = np.column_stack([xy, 0.05 * np.sin(3.0 * xy[:, 0]) * np.cos(2.0 * xy[:, 1]) + rng.normal(0, 0.002, n)]) + np.array([50.0, -20.0, 2.0])
    est = gt[rng.choice(n, 400_000, replace=False)].copy()
    est += np.array([0.012, -0.008, 0.015])                 # drift: a few fine cells
    est += rng.normal(0, 0.004, est.shape)
    est[:3000] += rng.normal(0, 0.08, (3000, 3))            # coarser levels
    est[3000:3400] += rng.normal(0, 1.5, (400, 3))          # octree
    with Engine(0) as eng:
        eng.upload(0, est, cell_size=0.1)
        eng.upload(1, gt, cell_size=0.1)
        eng.timers_enable(True)
        eng.timers_reset()
        idx, d2 = eng.nn1(0, 1)
        passes = eng.timer("nn_grid2")[1]
        eng.timers_enable(False)
        idx_b, d2_b = eng.nn1(1, 0)
    assert passes >= 1, "the cascade did not run: the scene is not dense enough for a fine grid below the radius grid"
    oi, od2 = oracle.nn1(gt, est)
    assert np.array_equal(d2, od2) and np.array_equal(idx, oi)
    oi, od2 = oracle.nn1(est, gt)
    assert np.array_equal(d2_b, od2) and np.array_equal(idx_b, oi)


def test_borrowed_device_input_gives_the_same_results_and_leaves_the_callers_buffer_alone(campus):
    """ME_FLAG_BORROW_DEVICE_INPUT (round 4): a device-resident cloud without a transform is read in place.  Same indices,
    distances, entropies and voxel tables as the copying context, bit for bit; a later transform of the cloud must not write
    into the caller's tensor."""
    import torch

    from cloud_map_evaluation_amd.engine import Engine

    est, gt = campus
    dev = torch.device("cuda", 0)
    e_d, g_d = torch.from_numpy(est).to(dev), torch.from_numpy(gt).to(dev)
    e_keep = e_d.clone()
    out = []
    for borrow in (False, True):
        with Engine(0, borrow_device_input=borrow) as eng:
            eng.upload(0, e_d, cell_size=0.1)
            eng.upload(1, g_d, cell_size=0.1)
            idx, d2 = eng.nn1(0, 1)
            m = eng.mme(0, 0.1, 10)
            tab = eng.voxel_gaussians(1, 1.0)
            T = np.eye(4)
            T[:3, 3] = [0.01, 0.02, -0.01]
            eng.transform_cloud(0, T)                # makes the cloud own its points first
            idx2, d22 = eng.nn1(0, 1)
            out.append((idx, d2, m, tab, idx2, d22))
        torch.cuda.synchronize()
        assert torch.equal(e_d, e_keep), "the engine wrote into a borrowed buffer"
    a, b = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


def test_index_rebuilds_with_the_pair_sort_when_the_packed_sort_had_to_cut_its_depth():
    """Round 4: the keys-only sort packs (key, point index) into 64 bits and lowers the sort depth when they do not fit; a DENSE
    cloud whose 1-NN grid then lands on the finest sorted level is indexed again with the pair sort at the full depth
    (me_index.hip).  A dense 3 m patch plus two points 600 m away (extent -> 13 bits of cells, 20 bits of index: depth 2 needs
    65 bits): two sorts must have run for that upload, one for the next upload of the slot, and the results are the oracle's."""
    import oracle

    from cloud_map_evaluation_amd.engine import Engine

    rng = np.random.default_rng(43)
    n = 700_000
    xy = rng.uniform(0.0, 3.0, (n, 2))
    gt = np.column_stack([xy, 0.04 * np.sin(4.0 * xy[:, 0]) + rng.normal(0, 0.002, n)])
    gt = np.concatenate([gt, np.array([[600.0, 10.0, 1.0], [-5.0, 610.0, 2.0]])])
    assert 2 ** 19 < len(gt) <= 2 ** 20
    est = gt[rng.choice(n, 200_000, replace=False)] + np.array([0.01, -0.01, 0.012]) + rng.normal(0, 0.004, (200_000, 3))
    with Engine(0) as eng:
        eng.timers_enable(True)
        eng.timers_reset()
        eng.upload(1, gt, cell_size=0.1)
        first = eng.timer("sort")[1]
        eng.timers_reset()
        eng.upload(1, gt, cell_size=0.1)
        second = eng.timer("sort")[1]
        eng.timers_reset()
        eng.upload(1, gt[: 2 ** 19 - 7], cell_size=0.1)  # ANOTHER cloud on the slot: the hint does not stick to it
        third = eng.timer("sort")[1]
        eng.timers_reset()
        eng.upload(1, gt, cell_size=0.1)
        fourth = eng.timer("sort")[1]
        eng.timers_enable(False)
        eng.upload(0, est, cell_size=0.1)
        idx, d2 = eng.nn1(0, 1)
        m = eng.mme(1, 0.1, 5)
    assert (first, second) == (2, 1), (first, second)
    assert third == 1 and fourth == 2, (third, fourth)  # (a sparse-enough prefix sorts once, packed; the dense cloud decides again)
    oi, od2 = oracle.nn1(gt, est)
    assert np.array_equal(d2, od2) and np.array_equal(idx, oi)
    om = oracle.mme(gt, 0.1, 5)
    assert m[3] == om[3] and np.array_equal(np.asarray(m[2]).astype(bool), om[2].astype(bool))  # valid counts and flags
    np.testing.assert_allclose(m[0], om[0], rtol=1e-9)
