"""Point-to-point ICP on the device-side correspondence / reduction step.

Restates the loop of Open3D's pipelines::registration::RegistrationICP with TransformationEstimationPointToPoint and the
default ICPConvergenceCriteria (relative_fitness = relative_rmse = 1e-6, max_iteration = 30) [upstream], which the
reference calls at map_eval.cpp:1369-1371 (registration_methods: 0).  Per iteration the GPU does the 1-NN search with the
`d2 < max^2` gate (me_nn1) and the sums Kabsch needs (me_icp_p2p_sums); the 3x3 solve below runs on the host;
me_transform_cloud applies the update.  Point-to-plane (1) and GICP (2) are not provided.
"""
from __future__ import annotations

import numpy as np

from .engine import ME_SLOT_EST, ME_SLOT_GT


def kabsch_update(n: int, sum_p, sum_q, sum_pq) -> np.ndarray:
    """Rigid update minimising sum |R p + t - q|^2 from the raw sums (Eigen::umeyama without scaling)."""
    sum_p, sum_q = np.asarray(sum_p, float), np.asarray(sum_q, float)
    pbar, qbar = sum_p / n, sum_q / n
    H = (np.asarray(sum_pq, float).reshape(3, 3) - n * np.outer(pbar, qbar)) / n  # cov(p, q) = E[(p-pbar)(q-qbar)^T]
    U, _, Vt = np.linalg.svd(H.T)  # sigma = E[(q-qbar)(p-pbar)^T] = H^T
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = qbar - R @ pbar
    return T


def _shift(T: np.ndarray, origin) -> np.ndarray:
    """The sums are relative to `origin`; express the update in absolute coordinates: x -> o + R (x - o) + t."""
    o = np.asarray(origin, float)
    A = np.eye(4)
    A[:3, :3] = T[:3, :3]
    A[:3, 3] = o + T[:3, 3] - T[:3, :3] @ o
    return A


def icp_point_to_point(eng, max_distance: float, max_iteration: int = 30, relative_fitness: float = 1e-6,
                       relative_rmse: float = 1e-6):
    """RegistrationICP(source = slot EST as uploaded (already moved by the initial matrix), target = slot GT).

    Returns dict(transformation (4x4 update accumulated on top of the uploaded state), fitness, inlier_rmse, n_corr,
    iterations).  On return slot EST holds the registered map (as `*map_3d_ = map_3d_->Transform(trans)`, :1392).
    """
    def evaluate():
        eng.nn1(ME_SLOT_EST, ME_SLOT_GT, fetch=False)
        s = eng.icp_p2p_sums(ME_SLOT_EST, max_distance)
        fit = s.n_corr / s.n_source if s.n_source else 0.0
        rmse = float(np.sqrt(s.sum_d2 / s.n_corr)) if s.n_corr else 0.0
        return s, fit, rmse

    total = np.eye(4)
    s, fit, rmse = evaluate()
    it = 0
    for it in range(1, max_iteration + 1):
        if s.n_corr < 3:
            break
        upd = _shift(kabsch_update(s.n_corr, list(s.sum_p), list(s.sum_q), list(s.sum_pq)), list(s.origin))
        total = upd @ total
        eng.transform_cloud(ME_SLOT_EST, upd)
        prev_fit, prev_rmse = fit, rmse
        s, fit, rmse = evaluate()
        if abs(prev_fit - fit) < relative_fitness and abs(prev_rmse - rmse) < relative_rmse:
            break
    return dict(transformation=total, fitness=fit, inlier_rmse=rmse, n_corr=int(s.n_corr), iterations=it)
