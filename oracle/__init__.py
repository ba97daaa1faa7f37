"""ctypes binding of the CPU oracle (oracle/mapeval_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (cloud_map_evaluation_amd) never imports this module.

Pinning status: AWD/CDF/SCS pinned by the reference's own run output (tests/golden/); KD-tree,
AC/COM/CD and MME are "parity unpinned" by the reference (it ships no tests and cannot be built here)
and are cross-checked against brute-force numpy / scipy.cKDTree in tests/test_oracle_*.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmapeval_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds). Returns the .so path."""
    src = os.path.join(_HERE, "mapeval_oracle.cpp")
    hdr = os.path.join(_HERE, "mapeval_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class _RegStats(C.Structure):
    _fields_ = [
        ("n_src", C.c_int64),
        ("n_corr", C.c_int64),
        ("number", C.c_double * 5),
        ("mean", C.c_double * 5),
        ("rmse", C.c_double * 5),
        ("fitness", C.c_double * 5),
        ("sigma", C.c_double * 5),
        ("sum_sqrt_all", C.c_double),
    ]


class _LsqSums(C.Structure):
    _fields_ = [("n_corr", C.c_int64), ("n_src", C.c_int64), ("JTJ", C.c_double * 36), ("JTr", C.c_double * 6),
                ("r2", C.c_double), ("sum_d2", C.c_double)]


@dataclass
class RegStats:
    n_src: int
    n_corr: int
    number: np.ndarray
    mean: np.ndarray
    rmse: np.ndarray
    fitness: np.ndarray
    sigma: np.ndarray
    sum_sqrt_all: float


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [dp, C.c_int64]
        L.orc_kdtree_build_mt.restype = C.c_void_p
        L.orc_kdtree_build_mt.argtypes = [dp, C.c_int64, C.c_int]
        L.orc_mme_points.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_double, C.c_int, dp,
                                     C.POINTER(C.c_uint8), C.c_int]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_nn1.argtypes = [C.c_void_p, dp, C.c_int64, ip, dp, C.c_int]
        L.orc_kdtree_radius_count.argtypes = [C.c_void_p, dp, C.c_int64, C.c_double, ip, C.c_int]
        L.orc_transform.argtypes = [dp, C.c_int64, dp]
        L.orc_voxel_downsample.restype = C.c_int64
        L.orc_voxel_downsample.argtypes = [dp, C.c_int64, C.c_double, dp, C.c_int64]
        L.orc_reg_stats_run.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_double, C.c_int, dp,
                                        C.POINTER(_RegStats), C.c_int]
        L.orc_chamfer.restype = C.c_double
        L.orc_chamfer.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_int]
        L.orc_mme.restype = C.c_double
        L.orc_mme.argtypes = [dp, C.c_int64, C.c_double, C.c_int, dp, C.POINTER(C.c_uint8),
                              C.POINTER(C.c_int64), dp, C.c_int, C.c_int]
        L.orc_voxel_build.restype = C.c_void_p
        L.orc_voxel_build.argtypes = [dp, C.c_int64, C.c_double]
        L.orc_voxel_free.argtypes = [C.c_void_p]
        L.orc_voxel_count.restype = C.c_int64
        L.orc_voxel_count.argtypes = [C.c_void_p]
        L.orc_voxel_export.argtypes = [C.c_void_p, ip, ip, dp, dp, dp]
        L.orc_w2_gaussian.restype = C.c_double
        L.orc_w2_gaussian.argtypes = [dp, dp, C.c_int, dp, dp, C.c_int]
        L.orc_awd_scs.restype = C.c_int
        L.orc_awd_scs.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, dp, dp,
                                  C.POINTER(C.c_int64), dp, dp, C.POINTER(C.c_int64)]
        L.orc_jet_color.argtypes = [C.c_double, dp]
        L.orc_render_distance.argtypes = [dp, C.c_int64, C.c_double, dp]
        L.orc_render_entropy.restype = C.c_int64
        L.orc_render_entropy.argtypes = [dp, dp, C.POINTER(C.c_uint8), C.c_int64, dp, dp, C.c_int64, dp, dp]
        L.orc_scs.restype = C.c_double
        L.orc_scs.argtypes = [ip, dp, C.c_int64, C.c_int]
        L.orc_kdtree_knn.argtypes = [C.c_void_p, dp, C.c_int64, C.c_int, ip, dp, C.c_int]
        L.orc_estimate_normals_knn.argtypes = [dp, C.c_int64, C.c_int, dp, C.c_int]
        L.orc_gicp_covariances.argtypes = [dp, C.c_int64, C.c_double, dp]
        L.orc_rotate_attributes.argtypes = [dp, dp, C.c_int64, dp]
        L.orc_icp_lsq_sums.argtypes = [C.c_int, dp, dp, C.c_int64, dp, dp, C.c_int64, C.c_double,
                                       C.POINTER(_LsqSums), C.c_int]
        _lib = L
    return _lib


def _pts(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("expected an (N,3) array")
    return a


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class KDTree:
    """A KD-tree kept alive across calls (KDTreeFlann::SetGeometry once, many searches): what the full-size parity tests
    and the CPU baseline use to query a 1 % subsample against the FULL tree.  build_threads = 1 is the reference's serial
    build; 0 = all cores (same tree, test infrastructure)."""

    def __init__(self, xyz, build_threads: int = 1):
        self.xyz = _pts(xyz)  # the tree points at this buffer
        self._t = lib().orc_kdtree_build_mt(_dp(self.xyz), self.xyz.shape[0], int(build_threads))

    def close(self):
        if getattr(self, "_t", None):
            lib().orc_kdtree_free(self._t)
            self._t = None

    __del__ = close

    def nn1(self, query, threads: int = 0):
        query = _pts(query)
        idx = np.empty(query.shape[0], np.int32)
        d2 = np.empty(query.shape[0], np.float64)
        lib().orc_kdtree_nn1(self._t, _dp(query), query.shape[0], _ip(idx), _dp(d2), threads)
        return idx, d2

    def radius_count(self, query, r: float, threads: int = 0):
        query = _pts(query)
        cnt = np.empty(query.shape[0], np.int32)
        lib().orc_kdtree_radius_count(self._t, _dp(query), query.shape[0], float(r), _ip(cnt), threads)
        return cnt

    def mme_points(self, sel, radius: float, min_k: int, threads: int = 0):
        """MME body (map_eval.cpp:1666-1701) for the tree's own points sel -> (entropies[m], valid[m] uint8)."""
        sel = np.ascontiguousarray(sel, dtype=np.int64)
        ent = np.zeros(sel.shape[0], np.float64)
        val = np.zeros(sel.shape[0], np.uint8)
        lib().orc_mme_points(self._t, sel.ctypes.data_as(C.POINTER(C.c_int64)), sel.shape[0], float(radius), int(min_k),
                             _dp(ent), val.ctypes.data_as(C.POINTER(C.c_uint8)), threads)
        return ent, val


def nn1(ref, query, threads: int = 0):
    """1-NN of every query in ref -> (idx int32[M], d2 float64[M])  (KDTreeFlann::SearchKNN k=1)."""
    ref, query = _pts(ref), _pts(query)
    t = lib().orc_kdtree_build(_dp(ref), ref.shape[0])
    idx = np.empty(query.shape[0], np.int32)
    d2 = np.empty(query.shape[0], np.float64)
    lib().orc_kdtree_nn1(t, _dp(query), query.shape[0], _ip(idx), _dp(d2), threads)
    lib().orc_kdtree_free(t)
    return idx, d2


def radius_count(ref, query, r: float, threads: int = 0):
    ref, query = _pts(ref), _pts(query)
    t = lib().orc_kdtree_build(_dp(ref), ref.shape[0])
    cnt = np.empty(query.shape[0], np.int32)
    lib().orc_kdtree_radius_count(t, _dp(query), query.shape[0], float(r), _ip(cnt), threads)
    lib().orc_kdtree_free(t)
    return cnt


def voxel_downsample(xyz, voxel_size: float) -> np.ndarray:
    """open3d VoxelDownSample (map_eval.cpp:38-39); output in ascending voxel-index order."""
    xyz = _pts(xyz)
    n = lib().orc_voxel_downsample(_dp(xyz), xyz.shape[0], float(voxel_size), None, 0)
    out = np.empty((n, 3), np.float64)
    lib().orc_voxel_downsample(_dp(xyz), xyz.shape[0], float(voxel_size), _dp(out), n)
    return out


def transform(xyz, T) -> np.ndarray:
    out = _pts(xyz).copy()
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    lib().orc_transform(_dp(out), out.shape[0], _dp(T))
    return out


def reg_stats(src, tgt, gate: float, gate_mode: int, trunc, threads: int = 1) -> RegStats:
    src, tgt = _pts(src), _pts(tgt)
    tr = np.ascontiguousarray(trunc, dtype=np.float64)
    assert tr.shape == (5,)
    out = _RegStats()
    lib().orc_reg_stats_run(_dp(src), src.shape[0], _dp(tgt), tgt.shape[0], float(gate), int(gate_mode),
                            _dp(tr), C.byref(out), threads)
    f = lambda x: np.array(list(x), dtype=np.float64)
    return RegStats(out.n_src, out.n_corr, f(out.number), f(out.mean), f(out.rmse), f(out.fitness),
                    f(out.sigma), out.sum_sqrt_all)


def chamfer(a, b, threads: int = 0) -> float:
    a, b = _pts(a), _pts(b)
    return lib().orc_chamfer(_dp(a), a.shape[0], _dp(b), b.shape[0], threads)


def mme(xyz, radius: float, min_k: int, mode: int = 2, threads: int = 0):
    """-> (mean_entropy, entropies[N], valid[N] uint8, n_valid, sum_entropy)."""
    xyz = _pts(xyz)
    n = xyz.shape[0]
    ent = np.zeros(n, np.float64)
    val = np.zeros(n, np.uint8)
    nv = C.c_int64(0)
    s = C.c_double(0.0)
    m = lib().orc_mme(_dp(xyz), n, float(radius), int(min_k), _dp(ent), val.ctypes.data_as(C.POINTER(C.c_uint8)),
                      C.byref(nv), C.byref(s), mode, threads)
    return m, ent, val, nv.value, s.value


class VoxelMap:
    def __init__(self, xyz, voxel_size: float):
        xyz = _pts(xyz)
        self.voxel_size = float(voxel_size)
        self._h = lib().orc_voxel_build(_dp(xyz), xyz.shape[0], self.voxel_size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_voxel_free(self._h)
            self._h = None

    def __len__(self):
        return lib().orc_voxel_count(self._h)

    def export(self):
        """-> keys[V,3] int32, n[V] int32, mu[V,3], sigma[V,3,3] (as stored), entropy[V]; ascending key order."""
        v = len(self)
        keys = np.empty((v, 3), np.int32)
        n = np.empty(v, np.int32)
        mu = np.empty((v, 3), np.float64)
        sg = np.empty((v, 9), np.float64)
        en = np.empty(v, np.float64)
        lib().orc_voxel_export(self._h, _ip(keys), _ip(n), _dp(mu), _dp(sg), _dp(en))
        return keys, n, mu, sg.reshape(v, 3, 3), en


def w2_gaussian(mu1, sigma1, n1, mu2, sigma2, n2) -> float:
    a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (mu1, sigma1, mu2, sigma2)]
    return lib().orc_w2_gaussian(_dp(a[0]), _dp(a[1]), int(n1), _dp(a[2]), _dp(a[3]), int(n2))


def awd_scs(gt: VoxelMap, est: VoxelMap, min_pts: int = 100, scs_radius: int = 5):
    """-> dict(awd, scs, rows[n,27], w_sorted[n], counts(active,old,new))."""
    cap = max(len(est), 1)
    rows = np.zeros((cap, 27), np.float64)
    ws = np.zeros(cap, np.float64)
    n = C.c_int64(cap)
    awd = C.c_double()
    scs = C.c_double()
    counts = (C.c_int64 * 3)()
    lib().orc_awd_scs(gt._h, est._h, est.voxel_size, min_pts, scs_radius, _dp(rows), _dp(ws), C.byref(n),
                      C.byref(awd), C.byref(scs), counts)
    k = n.value
    return dict(awd=awd.value, scs=scs.value, rows=rows[:k].copy(), w_sorted=ws[:k].copy(),
                counts=tuple(int(c) for c in counts))


def scs(keys, w, radius: int = 5) -> float:
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    return lib().orc_scs(_ip(keys), _dp(w), w.shape[0], radius)


def jet_color(value: float) -> np.ndarray:
    """open3d ColorMapJet.GetColor [upstream]."""
    out = np.empty(3, np.float64)
    lib().orc_jet_color(float(value), _dp(out))
    return out


def render_distance(d2, dis: float) -> np.ndarray:
    """renderDistanceOnPointCloud (map_eval.cpp:586-607): squared NN distances -> (N,3) Jet colours."""
    d2 = np.ascontiguousarray(d2, dtype=np.float64)
    out = np.empty((d2.shape[0], 3), np.float64)
    lib().orc_render_distance(_dp(d2), d2.shape[0], float(dis), _dp(out))
    return out


def render_entropy(xyz, entropies, valid):
    """ColorPointCloudByMME(pointcloud, entropies) (map_eval.cpp:686-735) -> (xyz_valid, rgb, min_abs, max_abs)."""
    xyz = _pts(xyz)
    ent = np.ascontiguousarray(entropies, dtype=np.float64)
    val = np.ascontiguousarray(valid, dtype=np.uint8)
    vp = val.ctypes.data_as(C.POINTER(C.c_uint8))
    mn, mx = C.c_double(), C.c_double()
    m = lib().orc_render_entropy(_dp(xyz), _dp(ent), vp, xyz.shape[0], None, None, 0, C.byref(mn), C.byref(mx))
    xo, co = np.empty((m, 3), np.float64), np.empty((m, 3), np.float64)
    lib().orc_render_entropy(_dp(xyz), _dp(ent), vp, xyz.shape[0], _dp(xo), _dp(co), m, C.byref(mn), C.byref(mx))
    return xo, co, mn.value, mx.value


# ---- registration_methods 1 / 2 (performICPRegistration, map_eval.cpp:1366-1394): Open3D pieces restated ----
def knn(ref, query, k: int, threads: int = 0):
    """k nearest neighbours, ascending by (d2, index) -> (idx int32[M,k], d2 float64[M,k])."""
    ref, query = _pts(ref), _pts(query)
    t = lib().orc_kdtree_build(_dp(ref), ref.shape[0])
    idx = np.empty((query.shape[0], k), np.int32)
    d2 = np.empty((query.shape[0], k), np.float64)
    lib().orc_kdtree_knn(t, _dp(query), query.shape[0], int(k), _ip(idx), _dp(d2), threads)
    lib().orc_kdtree_free(t)
    return idx, d2


def estimate_normals_knn(xyz, k: int = 20, threads: int = 0) -> np.ndarray:
    """open3d EstimateNormals(KDTreeSearchParamKNN(k)) on a cloud without normals."""
    xyz = _pts(xyz)
    out = np.empty_like(xyz)
    lib().orc_estimate_normals_knn(_dp(xyz), xyz.shape[0], int(k), _dp(out), threads)
    return out


def gicp_covariances(normals, epsilon: float = 1e-3) -> np.ndarray:
    """InitializePointCloudForGeneralizedICP -> (N,3,3)."""
    normals = _pts(normals)
    out = np.empty((normals.shape[0], 9), np.float64)
    lib().orc_gicp_covariances(_dp(normals), normals.shape[0], float(epsilon), _dp(out))
    return out.reshape(-1, 3, 3)


def rotate_attributes(T, normals=None, cov=None):
    """PointCloud::Transform on normals (N,3) / covariances (N,3,3); returns rotated copies."""
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    nn = None if normals is None else _pts(normals).copy()
    cc = None if cov is None else np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 9).copy()
    n = nn.shape[0] if nn is not None else cc.shape[0]
    lib().orc_rotate_attributes(None if nn is None else _dp(nn), None if cc is None else _dp(cc), n, _dp(T))
    return nn, (None if cc is None else cc.reshape(-1, 3, 3))


def icp_lsq_sums(mode: int, src, src_attr, tgt, tgt_attr, max_distance: float, threads: int = 0):
    """One linearised step: mode 1 point-to-plane (tgt_attr = normals), mode 2 generalized ICP (attrs = covariances).
    -> dict(n_corr, n_src, JTJ (6,6), JTr (6,), r2, sum_d2)."""
    src, tgt = _pts(src), _pts(tgt)
    width = 3 if mode == 1 else 9
    ta = np.ascontiguousarray(tgt_attr, dtype=np.float64).reshape(tgt.shape[0], width)
    sa = None if mode == 1 else np.ascontiguousarray(src_attr, dtype=np.float64).reshape(src.shape[0], 9)
    out = _LsqSums()
    lib().orc_icp_lsq_sums(int(mode), _dp(src), None if sa is None else _dp(sa), src.shape[0], _dp(tgt), _dp(ta),
                           tgt.shape[0], float(max_distance), C.byref(out), threads)
    return dict(n_corr=out.n_corr, n_src=out.n_src, JTJ=np.array(list(out.JTJ)).reshape(6, 6),
                JTr=np.array(list(out.JTr)), r2=out.r2, sum_d2=out.sum_d2)


def vector6_to_matrix(x) -> np.ndarray:
    """open3d utility::TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x[3:6]."""
    a, b, g = float(x[0]), float(x[1]), float(x[2])
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = np.asarray(x[3:6], float)
    return T


def registration_icp(mode: int, src, tgt, max_distance: float, tgt_normals=None, knn_k: int = 20, epsilon: float = 1e-3,
                     max_iteration: int = 30, relative_fitness: float = 1e-6, relative_rmse: float = 1e-6):
    """RegistrationICP (mode 1, point-to-plane; needs tgt_normals) / RegistrationGeneralizedICP (mode 2) with the default
    ICPConvergenceCriteria, init = identity.  -> dict(transformation, fitness, inlier_rmse, n_corr, iterations, cloud)."""
    src, tgt = _pts(src).copy(), _pts(tgt)
    if mode == 2:
        cs = gicp_covariances(estimate_normals_knn(src, knn_k), epsilon)
        ct = gicp_covariances(estimate_normals_knn(tgt, knn_k), epsilon)
    else:
        cs, ct = None, _pts(tgt_normals)

    def evaluate():
        s = icp_lsq_sums(mode, src, cs, tgt, ct, max_distance)
        n = s["n_corr"]
        return s, (n / len(src) if len(src) else 0.0), (float(np.sqrt(s["sum_d2"] / n)) if n else 0.0)

    total = np.eye(4)
    s, fit, rmse = evaluate()
    it = 0
    for it in range(1, max_iteration + 1):
        if s["n_corr"] == 0:
            break
        upd = vector6_to_matrix(np.linalg.solve(s["JTJ"], -s["JTr"]))
        total = upd @ total
        src = transform(src, upd)
        if cs is not None:
            _, cs = rotate_attributes(upd, cov=cs)
        pf, pr = fit, rmse
        s, fit, rmse = evaluate()
        if abs(pf - fit) < relative_fitness and abs(pr - rmse) < relative_rmse:
            break
    return dict(transformation=total, fitness=fit, inlier_rmse=rmse, n_corr=int(s["n_corr"]), iterations=it, cloud=src)
