"""ICP registration on the device-side correspondence / reduction step.

Restates the loop of Open3D's pipelines::registration::RegistrationICP [upstream] with the default ICPConvergenceCriteria
(relative_fitness = relative_rmse = 1e-6, max_iteration = 30), which the reference calls at map_eval.cpp:1366-1394:
  registration_methods 0  TransformationEstimationPointToPoint   -> icp_point_to_point  (me_icp_p2p_sums + Kabsch)
  registration_methods 1  TransformationEstimationPointToPlane   -> icp_point_to_plane  (me_icp_lsq_sums, 6x6 solve)
  registration_methods 2  RegistrationGeneralizedICP             -> icp_generalized     (me_estimate_normals(20) +
                                                                    me_gicp_covariances(1e-3) + me_icp_lsq_sums)
Per iteration the GPU does the 1-NN search with the `d2 < max^2` gate (me_nn1) and the sums; the small solve below runs
on the host; me_transform_cloud applies the update (points, normals and covariances).
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


def vector6_to_matrix(x) -> np.ndarray:
    """open3d utility::TransformVector6dToMatrix4d: R = Rz(x[2]) Ry(x[1]) Rx(x[0]), t = x[3:6]."""
    a, b, g = float(x[0]), float(x[1]), float(x[2])
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = np.asarray(x[3:6], float)
    return T


def lsq_update(JTJ, JTr) -> np.ndarray:
    """utility::SolveJacobianSystemAndObtainExtrinsicMatrix: x = solve(JTJ, -JTr) -> 4x4; identity when singular."""
    A = np.asarray(JTJ, float).reshape(6, 6)
    b = -np.asarray(JTr, float).reshape(6)
    try:
        x = np.linalg.solve(A, b)
    except np.linalg.LinAlgError:
        return np.eye(4)
    return vector6_to_matrix(x) if np.all(np.isfinite(x)) else np.eye(4)


def _icp_lsq(eng, mode: int, max_distance: float, max_iteration: int, relative_fitness: float, relative_rmse: float):
    def evaluate():
        eng.nn1(ME_SLOT_EST, ME_SLOT_GT, fetch=False)
        s = eng.icp_lsq_sums(ME_SLOT_EST, mode, max_distance)
        fit = s.n_corr / s.n_source if s.n_source else 0.0
        rmse = float(np.sqrt(s.sum_d2 / s.n_corr)) if s.n_corr else 0.0
        return s, fit, rmse

    total = np.eye(4)
    s, fit, rmse = evaluate()
    it = 0
    for it in range(1, max_iteration + 1):
        if s.n_corr == 0:  # ComputeTransformation returns the identity on an empty correspondence set
            break
        upd = lsq_update(list(s.JTJ), list(s.JTr))
        total = upd @ total
        eng.transform_cloud(ME_SLOT_EST, upd)
        prev_fit, prev_rmse = fit, rmse
        s, fit, rmse = evaluate()
        if abs(prev_fit - fit) < relative_fitness and abs(prev_rmse - rmse) < relative_rmse:
            break
    return dict(transformation=total, fitness=fit, inlier_rmse=rmse, n_corr=int(s.n_corr), iterations=it)


def icp_point_to_plane(eng, max_distance: float, max_iteration: int = 30, relative_fitness: float = 1e-6,
                       relative_rmse: float = 1e-6):
    """RegistrationICP(.., TransformationEstimationPointToPlane) (registration_methods 1, map_eval.cpp:1373-1377).
    The target (slot GT) must carry normals (Engine.set_normals; Open3D refuses a target without them as well)."""
    return _icp_lsq(eng, 1, max_distance, max_iteration, relative_fitness, relative_rmse)


def icp_generalized(eng, max_distance: float, epsilon: float = 1e-3, max_iteration: int = 30,
                    relative_fitness: float = 1e-6, relative_rmse: float = 1e-6):
    """RegistrationGeneralizedICP (registration_methods 2, map_eval.cpp:1378-1384): covariances of both clouds from
    their normals (estimated from the 20 nearest neighbours where the cloud has none), then the ICP loop."""
    eng.gicp_covariances(ME_SLOT_EST, epsilon)
    eng.gicp_covariances(ME_SLOT_GT, epsilon)
    return _icp_lsq(eng, 2, max_distance, max_iteration, relative_fitness, relative_rmse)
