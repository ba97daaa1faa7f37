"""BASELINE.json's full size (50 M-point pair) through size-independent properties: the oracle cannot follow there, so
the checks are identities the reference's definitions imply — self-distance zero, exact invariance of every COUNT under
a permutation of the input order, population conservation, monotone inlier counts, agreement of the one-call suite
with the piecewise calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)


@pytest.fixture(scope="module")
def big():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import Engine

    dev = torch.device("cuda", 0)
    est, gt = synth.campus_pair(50_000_000, density=2500.0, seed=100, device=dev)
    eng = Engine(0)
    yield eng, est, gt
    eng.close()


def test_full_size_suite_properties(big):
    import torch

    from cloud_map_evaluation_amd.engine import ME_GATE_LE_UNSQUARED, Param

    eng, est, gt = big
    n_e, n_g = est.shape[0], gt.shape[0]
    assert n_g == 50_000_000 and 40_000_000 < n_e < n_g
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=3.0)
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    out = eng.run_suite(P)
    eg, ge = out.est_gt, out.gt_est
    # correspondences and inliers: counts bounded by the source size, monotone in the threshold
    assert 0 < eg.n_corr <= n_e == eg.n_src and 0 < ge.n_corr <= n_g == ge.n_src
    for st in (eg, ge):
        num = list(st.number)
        assert all(num[k] >= num[k + 1] for k in range(4)) and num[0] <= st.n_corr
        fit = list(st.fitness)
        np.testing.assert_allclose(fit, np.array(num) / st.n_src, rtol=0, atol=0)
        assert all(0.0 <= m <= t for m, t in zip(list(st.mean), TRUNC))  # mean over ALL correspondences of capped terms
    assert out.full_chamfer == eg.mean_nn_dist + ge.mean_nn_dist and out.full_chamfer > 0
    assert 0 < out.mme_est_valid <= n_e and 0 < out.mme_gt_valid <= n_g
    assert out.mme_est > out.mme_gt  # the noisy map has the higher (less negative) entropy
    assert out.n_w_voxels > 1000 and np.isfinite(out.awd) and np.isfinite(out.scs)

    # voxel populations conserve the points
    for slot, n in ((0, n_e), (1, n_g)):
        keys, npts, mu, sigma, ent = eng.voxel_gaussians(slot, 3.0)
        assert int(npts.astype(np.int64).sum()) == n and len(np.unique(keys, axis=0)) == len(keys)

    # the one-call suite equals the piecewise calls bit for bit
    eng.nn1(0, 1, fetch=False)
    st = eng.nn_stats(0, 1.0, ME_GATE_LE_UNSQUARED, TRUNC)
    assert st.n_corr == eg.n_corr and np.array_equal(st.number, eg.number) and np.array_equal(st.rmse, eg.rmse)
    m = eng.mme(0, 0.1, 10, per_point=False)
    assert m[3] == out.mme_est_valid and m[0] == out.mme_est

    # permutation of the input order: every COUNT is exactly invariant, every mean to rounding
    g = torch.Generator(device=est.device)
    g.manual_seed(7)
    perm = torch.randperm(n_e, device=est.device, generator=g)
    eng.upload(0, est[perm].contiguous(), cell_size=0.1)
    eng.nn1(0, 1, fetch=False)
    st_p = eng.nn_stats(0, 1.0, ME_GATE_LE_UNSQUARED, TRUNC)
    assert st_p.n_corr == eg.n_corr and np.array_equal(st_p.number, eg.number)
    np.testing.assert_allclose(st_p.rmse, eg.rmse, rtol=1e-12)
    np.testing.assert_allclose(st_p.mean_nn_dist, eg.mean_nn_dist, rtol=1e-12)
    m_p = eng.mme(0, 0.1, 10, per_point=False)
    assert m_p[3] == out.mme_est_valid
    np.testing.assert_allclose(m_p[0], out.mme_est, rtol=1e-12)


def test_full_size_self_distance_is_zero(big):
    from cloud_map_evaluation_amd.engine import ME_GATE_LE_UNSQUARED

    eng, est, gt = big
    eng.upload(0, gt, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    eng.nn1(0, 1, fetch=False)
    st = eng.nn_stats(0, 1.0, ME_GATE_LE_UNSQUARED, TRUNC)
    assert st.n_corr == gt.shape[0] and st.mean_nn_dist == 0.0
    assert np.all(st.rmse == 0.0) and np.all(st.fitness == 1.0) and np.all(st.number == gt.shape[0])
    assert eng.computeChamferDistance() == 0.0


def test_full_size_generalized_icp_properties(big):
    """registration_methods 2 at the full size: normals are unit vectors, the correspondence count of the least-squares
    step is the d2 < max^2 count of the statistics pass (bit-exact), the normal matrix is symmetric positive definite, and
    registering an exactly moved copy of the 50 M cloud recovers the motion."""
    import torch

    from cloud_map_evaluation_amd.engine import ME_GATE_LT_SQUARED
    from cloud_map_evaluation_amd.icp import vector6_to_matrix

    eng, est, gt = big
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    nrm = eng.estimate_normals(1, 20)
    ln = np.linalg.norm(nrm, axis=1)
    assert np.all(np.isfinite(nrm)) and np.abs(ln - 1.0).max() < 1e-9
    del nrm, ln
    eng.gicp_covariances(0, 1e-3)
    eng.gicp_covariances(1, 1e-3)
    eng.nn1(0, 1, fetch=False)
    s = eng.icp_lsq_sums(0, 2, 0.5)
    st = eng.nn_stats(0, 0.5, ME_GATE_LT_SQUARED, TRUNC)
    assert s.n_corr == st.n_corr and s.n_source == est.shape[0]
    JTJ = np.array(list(s.JTJ)).reshape(6, 6)
    assert np.array_equal(JTJ, JTJ.T) and np.linalg.eigvalsh(JTJ).min() > 0
    assert 0 < s.r2 < 1e3 * s.sum_d2  # weighted residuals: eigenvalues of (Ct + Cs)^-1 lie in [1/2, 1/(2 eps)]
    # an exact copy of the ground truth, moved by a known rigid motion
    T0 = vector6_to_matrix([0.0004, -0.0003, 0.0005, 0.03, -0.02, 0.04])
    moved = gt @ torch.as_tensor(T0[:3, :3].T.copy(), device=gt.device) + torch.as_tensor(T0[:3, 3].copy(), device=gt.device)
    eng.upload(0, moved, cell_size=0.1)
    del moved
    res = eng.performICPRegistration(1.0, method=2)
    assert res["fitness"] == 1.0 and res["n_corr"] == gt.shape[0]
    assert res["inlier_rmse"] < 1e-6
    assert np.abs(res["transformation"] @ T0 - np.eye(4)).max() < 1e-7


def test_knn_rows_against_the_oracle_at_5m():
    """5 M points: sampled rows of the device k-NN equal the oracle's KD-tree rows bit for bit; every row is sorted and
    starts with the point itself."""
    import torch

    import oracle
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import Engine

    _, gt = synth.campus_pair(5_000_000, density=2500.0, seed=21, device=torch.device("cuda", 0))
    with Engine(0) as eng:
        eng.upload(1, gt, cell_size=0.1)
        _, idx, d2 = eng.estimate_normals(1, 20, with_neighbours=True)
    g = gt.cpu().numpy()
    assert np.array_equal(idx[:, 0], np.arange(len(g), dtype=np.int32)) and np.all(d2[:, 0] == 0.0)
    assert np.all(np.diff(d2, axis=1) >= 0)
    sel = np.random.default_rng(0).choice(len(g), 3000, replace=False)
    oidx, od2 = oracle.knn(g, g[sel], 20)
    assert np.array_equal(idx[sel], oidx) and np.array_equal(d2[sel], od2)
