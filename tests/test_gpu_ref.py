"""Device vs `oracle/_ref` — the REFERENCE'S OWN map_eval.cpp / voxel_calculator.cpp (compiled unmodified from /root/reference
over the stand-in headers of oracle/ref_build/; the prebuilt library travels to the GPU box).  VERDICT round 2, "next 1 (ii)".

  100 k  (C1)  the body of MapEval::process() (map_eval.cpp:51-85) in one go: computeMME (TBB est loop k >= 10, serial GT loop
               k >= 5), calculateMetricsWithInitialMatrix, calculateVMD — against Engine on the same pair, with an initial_matrix.
  1 M          the three MME loops (:1438-1535, :1538-1606, :1608-1737) per point; calculateMetrics (ICP-path gate, :1147-1202).
  5 M    (C2)  AC / COM both directions (:1204-1260 + :1069-1145 on the intended pairs), CD (:1398-1431), voxel tables
               (voxel_calculator.cpp:21-56), AWD / CDF / SCS (:240-390), est-MME through the reference's TBB loop.
Counts bit-exact; floating point 1e-9 relative (the contract of north_star is 1e-5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)
RTOL = 1e-9
from tests._tol import SIGMA_TOL  # |dSigma| / max|Sigma| per voxel


def _need():
    import torch
    from oracle import ref

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not ref.available():
        pytest.skip("oracle/_ref/libmapeval_ref.so was not built (needs /root/reference at build time)")
    return torch.device("cuda", 0), ref


def _stats_match(dev, got: dict):
    assert np.array_equal(dev.number, got["number"]), "inlier counts differ from the reference"
    for k in ("mean", "rmse", "sigma"):
        np.testing.assert_allclose(getattr(dev, k), got[k], rtol=RTOL, err_msg=k)
    assert np.array_equal(dev.fitness, got["fitness"])


def _voxel_rows_match(eng_rows, file_rows):
    """voxel_errors.txt is written in hash order with 6 significant digits; the engine's rows come in ascending key order."""
    assert eng_rows.shape == file_rows.shape
    order = np.lexsort((file_rows[:, 2], file_rows[:, 1], file_rows[:, 0]))
    f = file_rows[order]
    assert np.array_equal(eng_rows[:, 10:12], f[:, 10:12]), "voxel populations differ from the reference"
    np.testing.assert_allclose(eng_rows, f, rtol=2e-5, atol=1e-12)


def test_c1_process_body_against_the_reference():
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import Engine, Param

    dev, ref = _need()
    est_t, gt_t = synth.cube_pair(100_000, seed=42)
    est, gt = est_t.numpy(), gt_t.numpy()
    a = 0.002
    T = np.array([[np.cos(a), -np.sin(a), 0, 0.004], [np.sin(a), np.cos(a), 0, -0.003], [0, 0, 1, 0.002], [0, 0, 0, 1.0]])
    r = ref.suite_initial(est, gt, ref.config(trunc=TRUNC, icp_max_distance=1.0, nn_radius=0.1, vmd_voxel_size=0.5, T=T))
    p = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=0.5, initial_matrix_=T)
    with Engine(0) as eng:
        # the reference runs computeMME BEFORE it transforms the map (map_eval.cpp:56 then :1206)
        eng.upload(0, est, cell_size=0.1)
        eng.upload(1, gt, cell_size=0.1)
        m_est = eng.mme(0, 0.1, 10)
        m_gt = eng.mme(1, 0.1, 5)
        np.testing.assert_allclose(m_est[0], r["mme_est"], rtol=RTOL)
        np.testing.assert_allclose(m_gt[0], r["mme_gt"], rtol=RTOL)
        assert np.array_equal(m_est[2].astype(bool), r["est_entropies"] != 0), "est MME valid flags"
        assert np.array_equal(m_gt[2].astype(bool), r["gt_entropies"] != 0), "gt MME valid flags"
        np.testing.assert_allclose(m_est[1], r["est_entropies"], rtol=RTOL, atol=0)
        np.testing.assert_allclose(m_gt[1], r["gt_entropies"], rtol=RTOL, atol=0)
        eng.transform_cloud(0, T)
        assert np.array_equal(eng.download(0), r["est_transformed"]), "Transform differs bit-wise from the reference's"
        est_gt, gt_est, cd_vec = eng.calculateMetricsWithInitialMatrix(p)
        _stats_match(est_gt, r["est_gt"])
        # gt -> est: the reference's own call reads swapped indices (SURVEY finding 4); the same reference function on the
        # intended pairs is the comparison
        idx, d2 = eng.nn1(1, 0)
        keep = d2 <= 1.0
        pairs = np.stack([np.nonzero(keep)[0], idx[keep]], 1)
        _stats_match(gt_est, ref.diff_reg_result(0, gt, r["est_transformed"], pairs, TRUNC))
        np.testing.assert_allclose(est_gt.rmse, r["est_gt"]["rmse"], rtol=RTOL)
        v = eng.calculateVMD(0.5)
        np.testing.assert_allclose(v["awd"], r["vmd"], rtol=RTOL)
        np.testing.assert_allclose(v["scs"], r["scs"], rtol=RTOL)
        _voxel_rows_match(v["rows"], r["files"]["voxel_errors.txt"])
        np.testing.assert_allclose(v["w_sorted"], r["files"]["voxel_wasserstein_cdf.txt"][:, 0], rtol=2e-5)


def test_1m_three_mme_loops_and_the_icp_path_gate():
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import ME_GATE_LT_SQUARED, Engine

    dev, ref = _need()
    est_d, gt_d = synth.scan_pair(1_000_000, density=1200.0, seed=31, device=dev)
    est, gt = est_d.cpu().numpy(), gt_d.cpu().numpy()
    with Engine(0) as eng:
        eng.upload(0, est_d, cell_size=0.1)
        eng.upload(1, gt_d, cell_size=0.1)
        for variant, min_k, slot, cloud in ((2, 10, 0, est), (1, 10, 1, gt), (0, 5, 1, gt)):
            mean, ent, valid = ref.mme(variant, cloud, 0.1)
            d_mean, d_ent, d_valid, d_n, _ = eng.mme(slot, 0.1, min_k)
            assert np.array_equal(d_valid.astype(bool), valid), f"MME valid flags, reference loop {variant}"
            assert 0.2 * len(cloud) < d_n < len(cloud)
            np.testing.assert_allclose(d_ent, ent, rtol=RTOL, atol=0)
            np.testing.assert_allclose(d_mean, mean, rtol=RTOL)
        r = ref.calculate_metrics(est, gt, ref.config(trunc=TRUNC, icp_max_distance=0.3))
        eng.nn1(0, 1, fetch=False)
        eg = eng.nn_stats(0, 0.3, ME_GATE_LT_SQUARED, TRUNC)
        eng.nn1(1, 0, fetch=False)
        ge = eng.nn_stats(1, 0.3, ME_GATE_LT_SQUARED, TRUNC)
        assert eg.n_corr == r["n_corr"]
        _stats_match(eg, r["est_gt"])
        for k in ("mean", "rmse", "sigma"):
            np.testing.assert_allclose(getattr(ge, k), r["gt_est"][k], rtol=RTOL)
        assert np.array_equal(ge.fitness, r["gt_est"]["fitness"])
        np.testing.assert_allclose(eg.rmse + ge.rmse, r["cd_vec"], rtol=RTOL)
        np.testing.assert_allclose(eng.computeChamferDistance(), r["full_chamfer_dist"], rtol=RTOL)


def test_c2_5m_pair_against_the_reference():
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import ME_GATE_LE_UNSQUARED, Engine

    dev, ref = _need()
    est_d, gt_d = synth.scan_pair(5_000_000, density=2500.0, seed=100, device=dev)
    est, gt = est_d.cpu().numpy(), gt_d.cpu().numpy()
    # the reference's process() body without MME (C2 is CD + AC + COM; GT-MME is a serial loop): est -> gt statistics, VMD / SCS
    r = ref.suite_initial(est, gt, ref.config(trunc=TRUNC, icp_max_distance=1.0, vmd_voxel_size=3.0, evaluate_mme=False))
    with Engine(0) as eng:
        eng.upload(0, est_d, cell_size=0.1)
        eng.upload(1, gt_d, cell_size=0.1)
        eng.nn1(0, 1, fetch=False)
        _stats_match(eng.nn_stats(0, 1.0, ME_GATE_LE_UNSQUARED, TRUNC), r["est_gt"])
        idx, d2 = eng.nn1(1, 0)
        keep = d2 <= 1.0
        pairs = np.stack([np.nonzero(keep)[0], idx[keep]], 1)
        _stats_match(eng.nn_stats(1, 1.0, ME_GATE_LE_UNSQUARED, TRUNC), ref.diff_reg_result(0, gt, est, pairs, TRUNC))
        np.testing.assert_allclose(eng.computeChamferDistance(), ref.chamfer(est, gt), rtol=RTOL)
        v = eng.calculateVMD(3.0)
        assert v["n_rows"] > 300
        np.testing.assert_allclose(v["awd"], r["vmd"], rtol=RTOL)
        np.testing.assert_allclose(v["scs"], r["scs"], rtol=RTOL)
        _voxel_rows_match(v["rows"], r["files"]["voxel_errors.txt"])
        # the voxel tables of the whole clouds at full precision
        for slot, cloud in ((0, est), (1, gt)):
            e = ref.VoxelMap(cloud, 3.0).export()
            keys, n, mu, sg, en = eng.voxel_gaussians(slot, 3.0)
            assert np.array_equal(keys, e["keys"]) and np.array_equal(n, e["npts"])
            np.testing.assert_allclose(mu, e["mu"], rtol=1e-12, atol=1e-12)
            # Sigma: two-pass on the device, streaming Welford in the reference — compared against each matrix's own scale
            # (an off-diagonal entry may cancel to nothing); the entropy is a log-det of it
            scale = np.abs(e["sigma"]).max(axis=(1, 2), keepdims=True)
            assert np.max(np.abs(sg - e["sigma"]) / np.maximum(scale, 1e-300)) < SIGMA_TOL
            big = (n > 10) & (e["entropy"] != 0)
            np.testing.assert_allclose(en[big], e["entropy"][big], rtol=0, atol=1e-7)
        # est-MME of the whole 5 M cloud through the reference's TBB loop
        mean, ent, valid = ref.mme(2, est, 0.1)
        d_mean, d_ent, d_valid, d_n, _ = eng.mme(0, 0.1, 10)
        assert np.array_equal(d_valid.astype(bool), valid)
        np.testing.assert_allclose(d_ent, ent, rtol=RTOL, atol=0)
        np.testing.assert_allclose(d_mean, mean, rtol=RTOL)
