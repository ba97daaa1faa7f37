"""Oracle AC/COM/CD/MME legs vs independent numpy restatements (small sizes)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle
from cloud_map_evaluation_amd import synth

TRUNC = np.array([0.2, 0.1, 0.08, 0.05, 0.01])


def np_reg_stats(src, tgt, gate, gate_mode, trunc):
    """map_eval.cpp:1204-1260 + :1069-1145 written with numpy."""
    D = ((src[:, None, 0] - tgt[None, :, 0]) ** 2 + (src[:, None, 1] - tgt[None, :, 1]) ** 2) + (src[:, None, 2] - tgt[None, :, 2]) ** 2
    d2 = D.min(1)
    if gate < 0:
        keep = np.ones_like(d2, bool)
    elif gate_mode == 0:
        keep = d2 <= gate
    else:
        keep = d2 < gate * gate
    d = np.sqrt(d2[keep])
    C = d.size
    out = {"n_corr": C, "number": [], "mean": [], "rmse": [], "fitness": [], "sigma": []}
    for t in trunc:
        inl = d <= t
        mean = d[inl].sum() / C
        out["number"].append(inl.sum())
        out["mean"].append(mean)
        out["rmse"].append(np.sqrt((d[inl] ** 2).sum() / C))
        out["fitness"].append(inl.sum() / src.shape[0])
        out["sigma"].append(np.sqrt(((d - mean) ** 2).sum() / C))
    out["sum_sqrt_all"] = np.sqrt(d2).sum()
    return out


@pytest.mark.parametrize("gate,mode", [(1.0, 0), (0.01, 0), (0.15, 1), (-1.0, 0)])
def test_reg_stats_vs_numpy(gate, mode):
    est, gt = synth.cube_pair(3000, seed=42)
    est, gt = est.numpy(), gt.numpy()[:2500]
    got = oracle.reg_stats(est, gt, gate, mode, TRUNC)
    exp = np_reg_stats(est, gt, gate, mode, TRUNC)
    assert got.n_corr == exp["n_corr"] and got.n_src == est.shape[0]
    assert np.array_equal(got.number, np.array(exp["number"], float))  # bit-exact inlier counts
    for k in ("mean", "rmse", "fitness", "sigma"):
        np.testing.assert_allclose(getattr(got, k), exp[k], rtol=1e-12)
    np.testing.assert_allclose(got.sum_sqrt_all, exp["sum_sqrt_all"], rtol=1e-12)


def test_reg_stats_squared_vs_unsquared_gate_is_reproduced():
    # d = 0.9 m: d2 = 0.81 <= icp_max_distance 0.85 -> kept by the reference's gate (map_eval.cpp:1219)
    # although 0.9 > 0.85; gate_mode 1 (Open3D semantics) rejects it.
    src = np.array([[0.0, 0, 0], [10.0, 0, 0]])
    tgt = np.array([[0.9, 0, 0], [10.0, 0.95, 0]])
    assert oracle.reg_stats(src, tgt, 0.85, 0, TRUNC).n_corr == 1
    assert oracle.reg_stats(src, tgt, 0.85, 1, TRUNC).n_corr == 0


def test_reg_stats_empty_correspondences_is_nan():
    src = np.array([[0.0, 0, 0]])
    tgt = np.array([[5.0, 0, 0]])
    s = oracle.reg_stats(src, tgt, 1.0, 0, TRUNC)
    assert s.n_corr == 0 and np.all(np.isnan(s.mean)) and np.all(np.isnan(s.rmse)) and np.all(s.fitness == 0)


def test_chamfer_vs_scipy():
    est, gt = synth.cube_pair(4000, seed=1)
    est, gt = est.numpy(), gt.numpy()[:3000]
    cd = oracle.chamfer(est, gt)
    exp = cKDTree(gt).query(est)[0].mean() + cKDTree(est).query(gt)[0].mean()
    np.testing.assert_allclose(cd, exp, rtol=1e-12)
    a = oracle.reg_stats(est, gt, -1.0, 0, TRUNC).sum_sqrt_all / est.shape[0]
    b = oracle.reg_stats(gt, est, -1.0, 0, TRUNC).sum_sqrt_all / gt.shape[0]
    np.testing.assert_allclose(a + b, cd, rtol=1e-12)


def np_mme(p, r, min_k):
    """map_eval.cpp:1666-1701 with numpy (two-pass covariance, /(k-1), 0.5*log(2*pi*e*det))."""
    tree = cKDTree(p)
    ent = np.zeros(len(p))
    val = np.zeros(len(p), bool)
    for i, q in enumerate(p):
        d2 = ((p - q) ** 2)
        d2 = (d2[:, 0] + d2[:, 1]) + d2[:, 2]
        nb = np.nonzero(d2 < r * r)[0]
        order = np.argsort(d2[nb], kind="stable")
        nb = nb[order][1:]
        if len(nb) < min_k:
            continue
        pts = p[nb]
        c = pts - pts.mean(0)
        cov = c.T @ c / (len(nb) - 1)
        with np.errstate(all="ignore"):
            h = 0.5 * np.log(2 * np.pi * np.e * np.linalg.det(cov))
        if np.isfinite(h):
            ent[i] = h
            val[i] = True
    return ent, val


@pytest.mark.parametrize("min_k,mode", [(10, 2), (5, 0), (10, 1)])
def test_mme_vs_numpy(min_k, mode):
    est, _ = synth.cube_pair(20000, seed=9)
    p = est.numpy()[:2500] * 0.3  # denser
    mean, ent, val, nv, s = oracle.mme(p, 0.1, min_k, mode=mode)
    e2, v2 = np_mme(p, 0.1, min_k)
    assert np.array_equal(val.astype(bool), v2)  # bit-exact validity
    assert nv == v2.sum() and nv > 100
    np.testing.assert_allclose(ent, e2, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(mean, e2[v2].mean(), rtol=1e-10)
    np.testing.assert_allclose(s, e2[v2].sum(), rtol=1e-10)


def test_mme_modes_agree_and_no_valid_points():
    est, _ = synth.cube_pair(3000, seed=2)
    p = est.numpy()
    r0 = oracle.mme(p, 0.1, 10, mode=0)
    r1 = oracle.mme(p, 0.1, 10, mode=1)
    r2 = oracle.mme(p, 0.1, 10, mode=2)
    assert np.array_equal(r0[1], r1[1]) and np.array_equal(r0[1], r2[1])
    assert r0[3] == r1[3] == r2[3]
    sparse = oracle.mme(p[:50] * 100, 0.1, 10)
    assert sparse[0] == 0.0 and sparse[3] == 0  # returns 0 when nothing is valid (map_eval.cpp:1720-1724)


# ---- renderers (map_eval.cpp:586-607, :686-735; Open3D ColorMapJet) ----
def test_jet_colormap_known_values():
    """ColorMapJet [Open3D]: dark blue -> cyan -> green-ish -> yellow -> dark red; piecewise linear, clamped."""
    np.testing.assert_allclose(oracle.jet_color(0.0), [0.0, 0.0, 0.5])
    np.testing.assert_allclose(oracle.jet_color(0.125), [0.0, 0.0, 1.0])
    np.testing.assert_allclose(oracle.jet_color(0.375), [0.0, 1.0, 1.0])
    np.testing.assert_allclose(oracle.jet_color(0.5), [0.5, 1.0, 0.5])
    np.testing.assert_allclose(oracle.jet_color(0.625), [1.0, 1.0, 0.0])
    np.testing.assert_allclose(oracle.jet_color(0.875), [1.0, 0.0, 0.0])
    np.testing.assert_allclose(oracle.jet_color(1.0), [0.5, 0.0, 0.0])
    np.testing.assert_allclose(oracle.jet_color(-3.0), [0.0, 0.0, 0.0])
    np.testing.assert_allclose(oracle.jet_color(7.0), [0.0, 0.0, 0.0])


def test_render_distance_clamps_squared_distance_against_unsquared_threshold():
    d2 = np.array([0.0, 0.05, 0.2, 0.3, 4.0])
    rgb = oracle.render_distance(d2, 0.2)
    for i, v in enumerate([0.0, 0.25, 1.0, 1.0, 1.0]):  # min(d2, dis) / dis (map_eval.cpp:591-603)
        np.testing.assert_allclose(rgb[i], oracle.jet_color(v))


def test_render_entropy_range_and_compaction():
    rng = np.random.default_rng(5)
    xyz = rng.normal(size=(50, 3))
    ent = -rng.uniform(5.0, 9.0, 50)
    valid = rng.random(50) < 0.6
    ent[~valid] = 0.0
    xo, co, mn, mx = oracle.render_entropy(xyz, ent, valid)
    assert len(xo) == valid.sum() and np.array_equal(xo, xyz[valid])
    assert mn == abs(ent[valid].max()) and mx == abs(ent[valid].min())  # (:698-699)
    ne = (np.abs(ent[valid]) - mn) / (mx - mn)
    ne = (np.log(ne + 0.1) - np.log(0.1)) / (np.log(1.1) - np.log(0.1))
    for i in range(len(xo)):
        np.testing.assert_allclose(co[i], oracle.jet_color(ne[i]), atol=1e-12)
