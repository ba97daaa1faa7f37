"""Oracle KD-tree (nanoflann restatement) vs brute force and scipy.cKDTree.

The reference pins nothing here (no tests upstream) -> these cross-checks are what makes the oracle trustworthy
for the 1-NN / radius legs (map_eval.cpp:1218, :1415, :1670).
"""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle


def brute_d2(q, ref):
    dx = q[:, None, 0] - ref[None, :, 0]
    dy = q[:, None, 1] - ref[None, :, 1]
    dz = q[:, None, 2] - ref[None, :, 2]
    return (dx * dx + dy * dy) + dz * dz  # same association as nanoflann's L2 adaptor


@pytest.mark.parametrize("seed,n,m,scale", [(0, 5000, 2000, 10.0), (1, 17, 300, 1.0), (2, 1, 10, 1.0), (3, 4000, 1500, 1000.0)])
def test_nn1_bit_exact_vs_brute(seed, n, m, scale):
    rng = np.random.default_rng(seed)
    ref = rng.uniform(0, scale, (n, 3))
    q = rng.uniform(-0.1 * scale, 1.1 * scale, (m, 3))
    idx, d2 = oracle.nn1(ref, q)
    D = brute_d2(q, ref)
    assert np.array_equal(d2, D.min(1))          # bit-exact squared distances
    assert np.array_equal(D[np.arange(m), idx], d2)
    d_sp, _ = cKDTree(ref).query(q)
    np.testing.assert_allclose(np.sqrt(d2), d_sp, rtol=1e-12, atol=0)


def test_nn1_duplicates_and_clusters():
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 1, (200, 3))
    ref = np.concatenate([base, base, base[:50], np.zeros((40, 3))])  # exact duplicates, coincident block
    q = np.concatenate([base[:100], rng.uniform(-1, 2, (300, 3))])
    idx, d2 = oracle.nn1(ref, q)
    D = brute_d2(q, ref)
    assert np.array_equal(d2, D.min(1))
    assert np.all(d2[:100] == 0.0)
    # ties resolve to the smallest index
    assert np.array_equal(idx, D.argmin(1))


def test_nn1_serial_equals_parallel():
    rng = np.random.default_rng(7)
    ref = rng.normal(0, 5, (3000, 3))
    q = rng.normal(0, 6, (2000, 3))
    i1, d1 = oracle.nn1(ref, q, threads=1)
    i2, d2 = oracle.nn1(ref, q, threads=0)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


@pytest.mark.parametrize("r", [0.05, 0.3, 2.5])
def test_radius_count_strict_inequality(r):
    rng = np.random.default_rng(11)
    ref = rng.uniform(0, 3, (3000, 3))
    q = ref[:500]
    cnt = oracle.radius_count(ref, q, r)
    D = brute_d2(q, ref)
    assert np.array_equal(cnt, (D < r * r).sum(1))  # d2 < r*r, strict (nanoflann RadiusResultSet)
    assert np.all(cnt >= 1)                          # the query itself is always returned


def test_radius_boundary_point_excluded():
    ref = np.array([[0.0, 0, 0], [0.5, 0, 0], [0.25, 0, 0]])
    cnt = oracle.radius_count(ref, ref[:1], 0.5)
    assert cnt[0] == 2  # d2 == r*r is NOT inside


def test_transform_identity_and_rigid():
    rng = np.random.default_rng(3)
    p = rng.uniform(-100, 100, (1000, 3))
    assert np.array_equal(oracle.transform(p, np.eye(4)), p)
    th = 0.3
    T = np.eye(4)
    T[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    T[:3, 3] = [1.0, -2.0, 0.5]
    out = oracle.transform(p, T)
    np.testing.assert_allclose(out, p @ T[:3, :3].T + T[:3, 3], rtol=1e-13, atol=1e-12)
