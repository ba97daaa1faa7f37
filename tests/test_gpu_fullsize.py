"""Device vs oracle AT THE BASELINE.json SIZES ("missing 2").

C2  5 M pair, CD + AC + COM (map_eval.cpp:1215-1236, :1398-1431): the WHOLE clouds through the oracle — squared distances and
    neighbour indices of both directions bit for bit, the statistics of both gates, the Chamfer distance.
C3 / C4  20 M / 50 M pairs, full suite (:1666-1701): the oracle builds its KD-tree over the FULL cloud (orc_kdtree_build_mt:
    the reference's tree, built by OpenMP tasks) and answers a seeded 1 % query subsample: per-point entropy and valid flag
    of me_mme for both min_k, 1-NN squared distance and index, bit for bit (entropies: 1e-9); the voxel tables of the
    whole clouds (keys and populations exact) and AWD / SCS.
dense  one 10^4 pts/m^2 scene (the reference's default downsample_size 0.01 gives that density: ~314 neighbours in 0.1 m).
Counts bit-exact, floating point within 1e-9 relative, at every size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)
RTOL = 1e-9
from tests._tol import SIGMA_TOL  # |dSigma| / max|Sigma| per voxel


def _gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _stats_equal(dev_st, orc_st):
    assert dev_st.n_corr == orc_st.n_corr and dev_st.n_src == orc_st.n_src
    assert np.array_equal(dev_st.number, orc_st.number), "inlier counts differ from the oracle"
    np.testing.assert_allclose(dev_st.mean, orc_st.mean, rtol=RTOL)
    np.testing.assert_allclose(dev_st.rmse, orc_st.rmse, rtol=RTOL)
    np.testing.assert_allclose(dev_st.sigma, orc_st.sigma, rtol=RTOL)
    assert np.array_equal(dev_st.fitness, orc_st.fitness)


def test_c2_5m_pair_cd_ac_com_against_the_whole_oracle():
    import oracle
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import ME_GATE_LE_UNSQUARED, ME_GATE_LT_SQUARED, Engine

    dev = _gpu()
    est_d, gt_d = synth.scan_pair(5_000_000, density=2500.0, seed=100, device=dev)
    est, gt = est_d.cpu().numpy(), gt_d.cpu().numpy()
    assert len(est) == len(gt) == 5_000_000
    with Engine(0) as eng:
        eng.upload(0, est_d, cell_size=0.1)
        eng.upload(1, gt_d, cell_size=0.1)
        trees = {1: oracle.KDTree(gt, 0), 0: oracle.KDTree(est, 0)}
        clouds = {0: est, 1: gt}
        for q, r in ((0, 1), (1, 0)):
            idx, d2 = eng.nn1(q, r)
            oidx, od2 = trees[r].nn1(clouds[q], threads=0)
            assert np.array_equal(d2, od2), "squared 1-NN distances differ from the oracle"
            assert np.array_equal(idx, oidx), "1-NN indices differ from the oracle"
            # calculateMetricsWithInitialMatrix's gate (d2 <= 1.0, :1219) and the ICP path's (d2 < 1.0^2, :1168)
            for gate, mode in ((1.0, ME_GATE_LE_UNSQUARED), (0.5, ME_GATE_LT_SQUARED)):
                _stats_equal(eng.nn_stats(q, gate, mode, TRUNC), oracle.reg_stats(clouds[q], clouds[r], gate, mode, TRUNC, threads=0))
        np.testing.assert_allclose(eng.computeChamferDistance(), oracle.chamfer(est, gt), rtol=RTOL)
        for t in trees.values():
            t.close()


def _full_tree_subsample_check(make_pair, n, voxel, radius=0.1, frac=0.01, seed=5, expect_cascade=False):
    import oracle
    from cloud_map_evaluation_amd.engine import Engine

    dev = _gpu()
    est_d, gt_d = make_pair(dev)
    assert gt_d.shape[0] == n and est_d.shape[0] == n
    est, gt = est_d.cpu().numpy(), gt_d.cpu().numpy()
    rng = np.random.default_rng(seed)
    with Engine(0) as eng:
        if expect_cascade:
            eng.timers_enable(True)
            eng.timers_reset()
        eng.upload(0, est_d, cell_size=radius)
        eng.upload(1, gt_d, cell_size=radius)
        del est_d, gt_d
        clouds = {0: est, 1: gt}
        trees = {0: oracle.KDTree(est, 0), 1: oracle.KDTree(gt, 0)}  # FULL-size trees
        for slot, min_k in ((0, 10), (1, 5)):  # est k >= 10 (:1675), gt k >= 5 (:1458)
            sel = np.sort(rng.choice(n, int(n * frac), replace=False))
            mean, ent, val, n_valid, _ = eng.mme(slot, radius, min_k)
            o_ent, o_val = trees[slot].mme_points(sel, radius, min_k)
            assert np.array_equal(val[sel], o_val), "MME valid flags differ from the oracle"
            np.testing.assert_allclose(ent[sel], o_ent, rtol=RTOL, atol=0)
            assert n_valid == int(val.sum()) and 0.5 * n < n_valid <= n
            np.testing.assert_allclose(mean, ent[val.astype(bool)].mean(), rtol=1e-12)
            del ent, val
            # 1-NN of the subsample against the FULL other cloud
            idx, d2 = eng.nn1(slot, 1 - slot)
            oidx, od2 = trees[1 - slot].nn1(clouds[slot][sel], threads=0)
            assert np.array_equal(d2[sel], od2) and np.array_equal(idx[sel], oidx)
            del idx, d2
        for t in trees.values():
            t.close()
        if expect_cascade:  # the 1-NN passes above went through the multi-level list passes (k_nn_grid<FROM_LIST>), not only the fine grid
            assert eng.timer("nn_grid2")[1] >= 2, "the dense pair was meant to exercise the 1-NN cascade"
            eng.timers_enable(False)
        # voxel tables of the WHOLE clouds + AWD / SCS (voxel_calculator.cpp:21-56, map_eval.cpp:240-390)
        og, oe = oracle.VoxelMap(gt, voxel), oracle.VoxelMap(est, voxel)
        for slot, om in ((0, oe), (1, og)):
            keys, npts, mu, sigma, ent = eng.voxel_gaussians(slot, voxel)
            ok, on, omu, osig, oent = om.export()
            assert np.array_equal(keys, ok) and np.array_equal(npts, on), "voxel keys / populations differ from the oracle"
            np.testing.assert_allclose(mu, omu, rtol=RTOL, atol=1e-12)
            # two-pass (device) vs streaming Welford (reference order): compared against each matrix's own scale — a
            # single entry may cancel to nothing, the matrix as a whole may not
            scale = np.maximum(np.abs(osig).max(axis=(1, 2), keepdims=True), 1e-300)
            assert np.max(np.abs(sigma - osig) / scale) < SIGMA_TOL
        v, ov = eng.calculateVMD(voxel), oracle.awd_scs(og, oe)
        assert v["n_rows"] == len(ov["rows"]) > 100 and v["counts"] == tuple(ov["counts"])
        np.testing.assert_allclose(v["awd"], ov["awd"], rtol=RTOL)
        np.testing.assert_allclose(v["scs"], ov["scs"], rtol=RTOL)
        np.testing.assert_allclose(v["w_sorted"], ov["w_sorted"], rtol=1e-8)
        return v, ov


def test_c3_20m_pair_full_suite_against_the_full_tree_oracle():
    from cloud_map_evaluation_amd import synth

    _full_tree_subsample_check(lambda dev: synth.scan_pair(20_000_000, density=2500.0, seed=100, device=dev), 20_000_000, 3.0)


def test_c4_50m_multisession_pair_full_suite_against_the_full_tree_oracle():
    from cloud_map_evaluation_amd import synth

    _full_tree_subsample_check(lambda dev: synth.multisession_pair(50_000_000, 3, density=2500.0, seed=100, device=dev),
                               50_000_000, 3.0)


def test_c4_dense_50m_pair_full_suite_against_the_full_tree_oracle():
    """The reference's DEFAULT regime at full size (missing 4): every shipped config down-samples at
    downsample_size 0.01 (config/config.yaml:66) ~ 10^4 pts/m^2, ~300 neighbours inside nn_radius.  The 50 M + 50 M multi-session
    pair at that density: MME of both clouds and both 1-NN directions on a seeded subsample against FULL-size oracle trees — the
    queries the fine 1-NN grid leaves open go through the list passes over the coarser levels (checked to have run) —, voxel tables
    of the whole clouds, AWD / CDF / SCS."""
    from cloud_map_evaluation_amd import synth

    _full_tree_subsample_check(lambda dev: synth.multisession_pair(50_000_000, 3, density=10_000.0, seed=100, device=dev),
                               50_000_000, 3.0, frac=0.004, expect_cascade=True)


def test_c5_100m_tunnel_pair_full_suite_against_the_full_tree_oracle():
    """BASELINE.json configs[4]: 100 M + 100 M degenerate pair (tunnel + flat field + staircase), vmd_voxel_size 2.0
    (config_geode.yaml:60): the eigen-clamp and the near-singular Choleskys of voxel_calculator.cpp:119-137 at full size."""
    from cloud_map_evaluation_amd import synth

    n = 100_000_000
    v, ov = _full_tree_subsample_check(
        lambda dev: synth.tunnel_pair(n, density=2500.0, seed=300, device=dev, equal_sizes=True), n, 2.0, frac=0.005)
    rows = ov["rows"]
    sig = rows[:, 12:18]
    full = np.stack([sig[:, 0], sig[:, 1], sig[:, 2], sig[:, 1], sig[:, 3], sig[:, 4], sig[:, 2], sig[:, 4], sig[:, 5]], 1).reshape(-1, 3, 3)
    lam = np.linalg.eigvalsh(full / (rows[:, 11] - 1)[:, None, None])  # the third division (:120)
    assert (lam[:, 0] < 1e-6).mean() > 0.5, "the scene is meant to sit on the 1e-6 eigenvalue clamp"
    np.testing.assert_allclose(v["rows"][:, 9], rows[:, 9], rtol=1e-8)  # per-voxel W


def test_dense_scene_1e4_pts_per_m2_against_the_whole_oracle():
    """~314 neighbours per query (the reference's default downsample_size 0.01): MME of both clouds over ALL points, both
    1-NN directions, voxel tables and AWD / SCS."""
    import oracle
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import ME_GATE_LE_UNSQUARED, Engine

    dev = _gpu()
    est_d, gt_d = synth.scan_pair(2_000_000, density=10_000.0, seed=31, device=dev)
    est, gt = est_d.cpu().numpy(), gt_d.cpu().numpy()
    with Engine(0) as eng:
        eng.upload(0, est_d, cell_size=0.1)
        eng.upload(1, gt_d, cell_size=0.1)
        for slot, cloud, min_k in ((0, est, 10), (1, gt, 5)):
            mean, ent, val, n_valid, s = eng.mme(slot, 0.1, min_k)
            o = oracle.mme(cloud, 0.1, min_k, mode=1, threads=0)
            assert n_valid == o[3] and np.array_equal(val, o[2])
            np.testing.assert_allclose(ent, o[1], rtol=RTOL, atol=0)
            np.testing.assert_allclose(mean, o[0], rtol=RTOL)
        k = oracle.radius_count(gt, gt[:20000], 0.1)
        assert 200 < k.mean() < 400  # the density the test is about (edges and poles included)
        for q, r, a, b in ((0, 1, est, gt), (1, 0, gt, est)):
            idx, d2 = eng.nn1(q, r)
            oidx, od2 = oracle.nn1(b, a)
            assert np.array_equal(d2, od2) and np.array_equal(idx, oidx)
            _stats_equal(eng.nn_stats(q, 1.0, ME_GATE_LE_UNSQUARED, TRUNC), oracle.reg_stats(a, b, 1.0, 0, TRUNC, threads=0))
        v, ov = eng.calculateVMD(1.0), oracle.awd_scs(oracle.VoxelMap(gt, 1.0), oracle.VoxelMap(est, 1.0))
        assert v["n_rows"] == len(ov["rows"]) > 10
        np.testing.assert_allclose(v["awd"], ov["awd"], rtol=RTOL)
        np.testing.assert_allclose(v["scs"], ov["scs"], rtol=RTOL)
