"""Host part of the point-to-point ICP step (no GPU): the closed-form rigid update from the correspondence sums."""
import numpy as np

from cloud_map_evaluation_amd.icp import _shift, kabsch_update


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def test_kabsch_recovers_rigid_motion_from_sums():
    rng = np.random.default_rng(0)
    p = rng.normal(size=(500, 3)) * [3.0, 2.0, 0.5] + [10.0, -4.0, 2.0]
    R, t = _rot([1, 2, 3], 0.7), np.array([0.3, -1.2, 2.5])
    q = p @ R.T + t
    T = kabsch_update(len(p), p.sum(0), q.sum(0), p.T @ q)
    np.testing.assert_allclose(T[:3, :3], R, atol=1e-12)
    np.testing.assert_allclose(T[:3, 3], t, atol=1e-10)
    np.testing.assert_allclose(T[3], [0, 0, 0, 1])


def test_kabsch_never_returns_a_reflection():
    rng = np.random.default_rng(1)
    p = rng.normal(size=(200, 3))
    q = p * [1.0, 1.0, -1.0]  # mirrored target: best proper rotation, det = +1
    T = kabsch_update(len(p), p.sum(0), q.sum(0), p.T @ q)
    assert np.isclose(np.linalg.det(T[:3, :3]), 1.0)
    np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)


def test_kabsch_is_least_squares_optimal_under_noise():
    rng = np.random.default_rng(2)
    p = rng.normal(size=(2000, 3))
    R, t = _rot([0, 0, 1], 0.05), np.array([0.01, 0.02, -0.03])
    q = p @ R.T + t + rng.normal(scale=0.01, size=p.shape)
    T = kabsch_update(len(p), p.sum(0), q.sum(0), p.T @ q)

    def cost(A):
        return np.sum((p @ A[:3, :3].T + A[:3, 3] - q) ** 2)

    base = cost(T)
    for k in range(20):
        dT = np.eye(4)
        dT[:3, :3] = _rot(rng.normal(size=3), 1e-3 * rng.normal())
        dT[:3, 3] = 1e-3 * rng.normal(size=3)
        assert cost(dT @ T) >= base


def test_shift_moves_the_update_to_absolute_coordinates():
    rng = np.random.default_rng(3)
    o = np.array([1000.0, -2000.0, 50.0])
    p = rng.normal(size=(100, 3)) + o
    R, t = _rot([1, 0, 1], 0.2), np.array([0.5, 0.25, -0.125])
    q = p @ R.T + t
    pr, qr = p - o, q - o
    A = _shift(kabsch_update(len(p), pr.sum(0), qr.sum(0), pr.T @ qr), o)
    np.testing.assert_allclose(p @ A[:3, :3].T + A[:3, 3], q, atol=1e-9)
