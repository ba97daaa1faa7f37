"""The oracle's restatement of the Open3D pieces behind registration_methods 1 / 2 (performICPRegistration,
map_eval.cpp:1366-1394) against independent numpy / scipy formulations: k-NN vs brute force, normals vs numpy.linalg.eigh,
GICP covariances vs I - (1 - eps) n n^T, the J^T J / J^T r sums vs the literal Open3D per-row form (with the matrix square
root), TransformVector6dToMatrix4d vs scipy's Euler angles, and convergence of the loops.  Parity with Open3D itself is
unpinned (no Open3D, no vectors in the reference tree)."""
import numpy as np
import pytest
import scipy.linalg
from scipy.spatial.transform import Rotation

import oracle
from cloud_map_evaluation_amd import icp, synth


def _pair(n=8000, seed=3):
    est, gt = synth.campus_pair(n, seed=seed)
    return est.numpy(), gt.numpy()


def test_knn_matches_brute_force_bit_for_bit():
    rng = np.random.default_rng(0)
    ref = rng.uniform(0, 10, (3000, 3))
    ref[100:110] = ref[100]  # exact duplicates: ties resolved by index
    q = np.vstack([rng.uniform(-1, 11, (150, 3)), ref[:50]])
    idx, d2 = oracle.knn(ref, q, 20)
    D = (q[:, None, :] - ref[None, :, :]) ** 2
    D = (D[:, :, 0] + D[:, :, 1]) + D[:, :, 2]
    order = np.lexsort((np.broadcast_to(np.arange(len(ref)), D.shape), D), axis=1)[:, :20]
    assert np.array_equal(idx, order)
    assert np.array_equal(d2, np.take_along_axis(D, order, 1))
    idx, d2 = oracle.knn(ref[:5], q[:3], 8)  # fewer points than k
    assert np.all(idx[:, 5:] == -1) and np.all(np.isinf(d2[:, 5:])) and np.all(idx[:, :5] >= 0)


def test_normals_are_the_smallest_eigenvector_of_the_neighbourhood_covariance():
    _, gt = _pair()
    nrm = oracle.estimate_normals_knn(gt, 20)
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-12
    idx, _ = oracle.knn(gt, gt, 20)
    worst = 0.0
    checked = 0
    for i in range(0, len(gt), 11):
        P = gt[idx[i]]
        w, v = np.linalg.eigh(np.cov(P.T, bias=True))
        if (w[1] - w[0]) < 1e-3 * w[2]:
            continue
        worst = max(worst, 1 - abs(v[:, 0] @ nrm[i]))
        checked += 1
    assert checked > 300 and worst < 1e-9


def test_normals_degenerate_inputs():
    # exact plane z = 1: normal (0,0,+-1); exact line: any unit vector orthogonal to it; < 3 points: (0,0,1)
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(5.0)), -1).reshape(-1, 2)
    plane = np.c_[g * np.array([0.37, 0.21]), np.ones(len(g))]
    n = oracle.estimate_normals_knn(plane, 9)
    assert np.allclose(np.abs(n), [0, 0, 1], atol=1e-12)
    line = np.c_[np.arange(10.0) * 0.3, np.zeros(10), np.zeros(10)]
    n = oracle.estimate_normals_knn(line, 5)
    assert np.allclose(np.linalg.norm(n, axis=1), 1) and np.abs(n[:, 0]).max() < 1e-12
    assert np.array_equal(oracle.estimate_normals_knn(np.array([[0.0, 0, 0], [1, 2, 3]]), 20), [[0, 0, 1], [0, 0, 1]])
    same = np.zeros((6, 3)) + 2.5  # all points coincide: zero covariance -> zero vector -> (0,0,1)
    assert np.array_equal(oracle.estimate_normals_knn(same, 4), np.tile([0.0, 0, 1], (6, 1)))


def test_gicp_covariances():
    rng = np.random.default_rng(2)
    n = rng.normal(size=(500, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[0] = [-1, 0, 0]          # (sic) nearly opposite to e1: identity rotation -> diag(eps, 1, 1)
    n[1] = [1, 0, 0]
    c = oracle.gicp_covariances(n, 1e-3)
    ref = np.eye(3)[None] - (1 - 1e-3) * n[:, :, None] * n[:, None, :]
    keep = n[:, 0] >= -0.99
    assert np.abs(c[keep] - ref[keep]).max() < 1e-12
    assert np.allclose(c[~keep], np.diag([1e-3, 1, 1]))
    assert np.abs(c - np.swapaxes(c, 1, 2)).max() < 1e-15
    T = icp.vector6_to_matrix([0.3, -0.2, 0.5, 1, 2, 3])
    n2, c2 = oracle.rotate_attributes(T, n, c)
    R = T[:3, :3]
    assert np.abs(n2 - n @ R.T).max() < 1e-15 and np.abs(c2 - R @ c @ R.T).max() < 1e-15


def test_vector6_to_matrix_is_rz_ry_rx():
    x = [0.11, -0.23, 0.37, 1.5, -2.5, 0.5]
    for T in (icp.vector6_to_matrix(x), oracle.vector6_to_matrix(x)):
        assert np.abs(T[:3, :3] - Rotation.from_euler("xyz", x[:3]).as_matrix()).max() < 1e-15  # extrinsic x, y, z
        assert np.array_equal(T[:3, 3], x[3:]) and np.array_equal(T[3], [0, 0, 0, 1])
    assert np.array_equal(icp.lsq_update(np.zeros((6, 6)), np.ones(6)), np.eye(4))  # singular system: identity


def test_lsq_sums_equal_the_literal_open3d_rows():
    est, gt = _pair(4000)
    n_gt = oracle.estimate_normals_knn(gt, 20)
    cs = oracle.gicp_covariances(oracle.estimate_normals_knn(est, 20))
    ct = oracle.gicp_covariances(n_gt)
    idx, d2 = oracle.nn1(gt, est)
    m = d2 < 0.25 ** 2
    # point-to-plane
    s = oracle.icp_lsq_sums(1, est, None, gt, n_gt, 0.25)
    J = np.hstack([np.cross(est[m], n_gt[idx[m]]), n_gt[idx[m]]])
    r = np.einsum("ij,ij->i", est[m] - gt[idx[m]], n_gt[idx[m]])
    assert s["n_corr"] == m.sum() and s["n_src"] == len(est)
    assert np.allclose(s["JTJ"], J.T @ J, rtol=1e-10) and np.allclose(s["JTr"], J.T @ r, rtol=1e-9, atol=1e-9)
    assert np.isclose(s["r2"], r @ r, rtol=1e-12) and np.isclose(s["sum_d2"], d2[m].sum(), rtol=1e-12)
    # generalized: W = (Ct + Cs)^(-1/2); three rows per correspondence
    s = oracle.icp_lsq_sums(2, est, cs, gt, ct, 0.25)
    JTJ, JTr, r2 = np.zeros((6, 6)), np.zeros(6), 0.0
    for i in np.nonzero(m)[0]:
        W = np.real(scipy.linalg.sqrtm(np.linalg.inv(ct[idx[i]] + cs[i])))
        x, y, z = est[i]
        Jm = W @ np.array([[0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1.0]])
        rr = W @ (est[i] - gt[idx[i]])
        JTJ += Jm.T @ Jm
        JTr += Jm.T @ rr
        r2 += rr @ rr
    assert np.allclose(s["JTJ"], JTJ, rtol=1e-8) and np.allclose(s["JTr"], JTr, rtol=1e-7, atol=1e-7)
    assert np.isclose(s["r2"], r2, rtol=1e-9)
    assert np.abs(np.linalg.solve(s["JTJ"], -s["JTr"]) - np.linalg.solve(JTJ, -JTr)).max() < 1e-9


@pytest.mark.parametrize("mode", [1, 2])
def test_registration_loops_converge(mode):
    _, gt = _pair(12000, seed=5)
    T0 = icp.vector6_to_matrix([0.004, -0.003, 0.006, 0.05, -0.04, 0.03])
    src = oracle.transform(gt, T0)
    nrm = oracle.estimate_normals_knn(gt, 20) if mode == 1 else None
    r = oracle.registration_icp(mode, src, gt, 1.0, tgt_normals=nrm)
    assert r["fitness"] == 1.0 and r["inlier_rmse"] < 1e-9 and r["iterations"] <= 10
    assert np.abs(r["transformation"] @ T0 - np.eye(4)).max() < 1e-9
