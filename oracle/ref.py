"""ctypes binding of oracle/_ref/libmapeval_ref.so — the reference's OWN map_eval.cpp + voxel_calculator.cpp.

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench's cpu_baseline leg).  The library is built by `oracle/ref_build/Makefile`
from the sources where they lie under /root/reference (nothing is copied into this repository), over the functional
stand-in headers in `oracle/ref_build/standin/` (Eigen, Open3D, TBB, PCL, yaml-cpp are absent from this image).  What runs
inside it is therefore the reference's control flow and expressions, on the builder's matrix / KD-tree primitives
(DESIGN.md section 2 lists them).

/root/reference does not exist on the GPU box: there only the prebuilt `oracle/_ref/libmapeval_ref.so` (it travels with the
snapshot, git-ignored) is used; `available()` says whether a library can be had.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "ref_build")
_SO = os.path.join(_HERE, "_ref", "libmapeval_ref.so")
_SO_PATCHED = os.path.join(_HERE, "_ref", "libmapeval_ref_patched.so")
REFERENCE = os.environ.get("MAPEVAL_REFERENCE", "/root/reference/map_eval")


def have_sources() -> bool:
    return os.path.exists(os.path.join(REFERENCE, "src", "map_eval.cpp"))


def build(force: bool = False) -> str | None:
    """(Re)build the library when the reference sources are present; return its path, or None when there is neither a
    prebuilt library nor sources."""
    if have_sources():
        subprocess.check_call(["make", "-C", _BUILD, "-s", f"REF={REFERENCE}"] + (["-B"] if force else []))
    return _SO if os.path.exists(_SO) else None


def available() -> bool:
    return os.path.exists(_SO) or have_sources()


def build_patched(force: bool = False) -> str | None:
    """The reference's own map_eval.cpp with the binding of INTEGRATION.md section B applied at build time
    (oracle/ref_build/apply_binding.py; needs libmapeval_hip.so): oracle/_ref/libmapeval_ref_patched.so, or None."""
    if have_sources():
        subprocess.check_call(["make", "-C", _BUILD, "-s", f"REF={REFERENCE}", "patched"] + (["-B"] if force else []))
    return _SO_PATCHED if os.path.exists(_SO_PATCHED) else None


def patched_available() -> bool:
    return os.path.exists(_SO_PATCHED) or have_sources()


_lib_patched = None


def process_patched(cfg: "Config", workdir, gt_path) -> dict:
    """MapEval::process() of the PATCHED reference (its three hot-path calls go to libmapeval_hip.so) on the same files as
    process().  rc = -1 when the library reports no GPU: the reference's own error path, no CPU fallback."""
    global _lib_patched
    if _lib_patched is None:
        so = build_patched()
        if so is None:
            raise RuntimeError("oracle/_ref/libmapeval_ref_patched.so is missing and /root/reference is not here to build it")
        L = C.CDLL(so)
        L.ref_last_error.restype = C.c_char_p
        L.ref_process.argtypes = [C.POINTER(Config), C.c_char_p, C.c_char_p, C.POINTER(Results), C.POINTER(C.c_int)]
        _lib_patched = L
    r = Results()
    rc = C.c_int(0)
    wd = str(workdir).rstrip("/") + "/"
    if _lib_patched.ref_process(C.byref(cfg), wd.encode(), str(gt_path).encode(), C.byref(r), C.byref(rc)) != 0:
        raise RuntimeError("reference (patched): " + _lib_patched.ref_last_error().decode())
    out = _results(r)
    out["rc"] = rc.value
    out["files"] = _read_outputs(wd) if rc.value == 0 else {}
    return out


class Config(C.Structure):
    _fields_ = [("trunc", C.c_double * 5), ("icp_max_distance", C.c_double), ("nn_radius", C.c_double),
                ("vmd_voxel_size", C.c_double), ("downsample_size", C.c_double), ("T", C.c_double * 16),
                ("evaluate_mme", C.c_int32), ("evaluate_gt_mme", C.c_int32), ("use_tbb_mme", C.c_int32),
                ("evaluate_using_initial", C.c_int32), ("save_immediate_result", C.c_int32),
                ("registration_methods", C.c_int32)]


class Results(C.Structure):
    _fields_ = [("n_est", C.c_int64), ("n_gt", C.c_int64), ("n_est_gt_vecs", C.c_int64), ("n_gt_est_vecs", C.c_int64),
                ("est_gt", (C.c_double * 5) * 5), ("gt_est", (C.c_double * 5) * 5),
                ("cd_vec", C.c_double * 5), ("f1_vec", C.c_double * 5), ("iou_vec", C.c_double * 5),
                ("mme_est", C.c_double), ("mme_gt", C.c_double), ("min_abs_entropy", C.c_double),
                ("max_abs_entropy", C.c_double), ("vmd", C.c_double), ("scs", C.c_double),
                ("full_chamfer_dist", C.c_double), ("trans", C.c_double * 16)]


ROWS = ("mean", "rmse", "fitness", "sigma", "number")  # order in which the reference pushes its Vector5d (:1140-1144)

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = build()
        if so is None:
            raise RuntimeError("oracle/_ref/libmapeval_ref.so is missing and /root/reference is not here to build it")
        L = C.CDLL(so)
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        i64p = C.POINTER(C.c_int64)
        L.ref_last_error.restype = C.c_char_p
        L.ref_suite_initial.argtypes = [dp, C.c_int64, dp, C.c_int64, C.POINTER(Config), C.c_char_p, C.POINTER(Results),
                                        dp, dp, dp]
        L.ref_process.argtypes = [C.POINTER(Config), C.c_char_p, C.c_char_p, C.POINTER(Results), C.POINTER(C.c_int)]
        L.ref_calculate_metrics.argtypes = [dp, C.c_int64, dp, C.c_int64, C.POINTER(Config), C.c_char_p,
                                            C.POINTER(Results), i64p]
        L.ref_diff_reg_result.argtypes = [C.c_int, dp, C.c_int64, dp, C.c_int64, ip, C.c_int64, dp, C.c_char_p,
                                          C.POINTER((C.c_double * 5) * 5), i64p]
        L.ref_chamfer.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_char_p, dp]
        L.ref_mme.argtypes = [C.c_int, dp, C.c_int64, C.c_double, C.c_char_p, dp, bp, dp]
        L.ref_mme_raw.argtypes = [C.c_int, dp, C.c_int64, C.c_double, C.c_char_p, dp, bp, bp, dp]
        L.ref_compute_entropy.restype = C.c_double
        L.ref_compute_entropy.argtypes = [dp]
        L.ref_vmd.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_double, C.c_char_p, dp, dp]
        L.ref_voxel_build.restype = C.c_void_p
        L.ref_voxel_build.argtypes = [dp, C.c_int64, C.c_double]
        L.ref_voxel_free.argtypes = [C.c_void_p]
        L.ref_voxel_count.restype = C.c_int64
        L.ref_voxel_count.argtypes = [C.c_void_p]
        L.ref_voxel_export.argtypes = [C.c_void_p, ip, ip, dp, dp, dp, dp, ip]
        L.ref_voxel_update.argtypes = [C.c_void_p, C.c_void_p, i64p]
        L.ref_w2_gaussian.restype = C.c_double
        L.ref_w2_gaussian.argtypes = [dp, dp, C.c_int, dp, dp, C.c_int]
        L.ref_voxel_index.argtypes = [dp, C.c_double, ip]
        L.ref_neighbor_indices.restype = C.c_int64
        L.ref_neighbor_indices.argtypes = [ip, C.c_int, ip]
        L.ref_kdtree_nn1.argtypes = [dp, C.c_int64, dp, C.c_int64, ip, dp]
        L.ref_kdtree_radius_count.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_double, ip]
        _lib = L
    return _lib


def _pts(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _check(rc):
    if rc != 0:
        raise RuntimeError("reference: " + lib().ref_last_error().decode())


class _Workdir:
    """The reference's MapEval constructor creates <dir>/map_results/ and opens map_results.txt there (map_eval.h:158-173)."""

    def __init__(self, path=None):
        self.own = path is None
        self.path = tempfile.mkdtemp(prefix="mapeval_ref_") if path is None else str(path)
        os.makedirs(self.path, exist_ok=True)

    def __enter__(self):
        return self.path.rstrip("/") + "/"

    def __exit__(self, *a):
        if self.own:
            shutil.rmtree(self.path, ignore_errors=True)


def config(trunc=(0.2, 0.1, 0.08, 0.05, 0.01), icp_max_distance=1.0, nn_radius=0.1, vmd_voxel_size=3.0,
           downsample_size=0.01, T=None, evaluate_mme=True, evaluate_gt_mme=True, use_tbb_mme=True,
           evaluate_using_initial=True, save_immediate_result=False, registration_methods=2) -> Config:
    c = Config()
    c.trunc[:] = [float(t) for t in trunc]
    c.icp_max_distance, c.nn_radius, c.vmd_voxel_size, c.downsample_size = icp_max_distance, nn_radius, vmd_voxel_size, downsample_size
    c.T[:] = [float(v) for v in (np.eye(4) if T is None else np.asarray(T, float)).reshape(-1)]
    c.evaluate_mme, c.evaluate_gt_mme, c.use_tbb_mme = int(evaluate_mme), int(evaluate_gt_mme), int(use_tbb_mme)
    c.evaluate_using_initial, c.save_immediate_result = int(evaluate_using_initial), int(save_immediate_result)
    c.registration_methods = registration_methods
    return c


def _results(r: Results) -> dict:
    def vecs(m, n):
        a = np.array([[m[i][k] for k in range(5)] for i in range(5)])
        return {ROWS[i]: a[i] for i in range(int(n))}

    return {"n_est": r.n_est, "n_gt": r.n_gt, "est_gt": vecs(r.est_gt, r.n_est_gt_vecs), "gt_est": vecs(r.gt_est, r.n_gt_est_vecs),
            "cd_vec": np.array(r.cd_vec[:]), "f1_vec": np.array(r.f1_vec[:]), "iou_vec": np.array(r.iou_vec[:]),
            "mme_est": r.mme_est, "mme_gt": r.mme_gt, "min_abs_entropy": r.min_abs_entropy, "max_abs_entropy": r.max_abs_entropy,
            "vmd": r.vmd, "scs": r.scs, "full_chamfer_dist": r.full_chamfer_dist, "trans": np.array(r.trans[:]).reshape(4, 4)}


def suite_initial(est, gt, cfg: Config, workdir=None) -> dict:
    """process() after loading (map_eval.cpp:51-85): computeMME, calculateMetricsWithInitialMatrix, calculateVMD."""
    est, gt = _pts(est), _pts(gt)
    r = Results()
    e_ent = np.zeros(len(est)) if cfg.evaluate_mme else None
    g_ent = np.zeros(len(gt)) if (cfg.evaluate_mme and cfg.evaluate_gt_mme) else None
    est_out = np.empty_like(est)
    with _Workdir(workdir) as wd:
        _check(lib().ref_suite_initial(_dp(est), len(est), _dp(gt), len(gt), C.byref(cfg), wd.encode(), C.byref(r), _dp(e_ent),
                                       _dp(g_ent), _dp(est_out)))
        out = _results(r)
        out["files"] = _read_outputs(wd)
    out.update(est_entropies=e_ent, gt_entropies=g_ent, est_transformed=est_out)
    return out


def process(cfg: Config, workdir, gt_path) -> dict:
    """MapEval::process() on <workdir>/global_pcd_lidar.pcd and gt_path (binary or ascii PCD, fp32/fp64 x y z)."""
    r = Results()
    rc = C.c_int(0)
    wd = str(workdir).rstrip("/") + "/"
    _check(lib().ref_process(C.byref(cfg), wd.encode(), str(gt_path).encode(), C.byref(r), C.byref(rc)))
    out = _results(r)
    out["rc"] = rc.value
    out["files"] = _read_outputs(wd)
    return out


def calculate_metrics(est, gt, cfg: Config) -> dict:
    """calculateMetrics (ICP path, :1147-1202) with the correspondence set of EvaluateRegistration(map, gt, max)."""
    est, gt = _pts(est), _pts(gt)
    r = Results()
    n = C.c_int64(0)
    with _Workdir() as wd:
        _check(lib().ref_calculate_metrics(_dp(est), len(est), _dp(gt), len(gt), C.byref(cfg), wd.encode(), C.byref(r), C.byref(n)))
    out = _results(r)
    out["n_corr"] = n.value
    return out


def diff_reg_result(variant: int, src, tgt, pairs, trunc) -> dict:
    """0: getDiffRegResultWithCorrespondence (:1069-1145); 1: 6-arg getDiffRegResult (:990-1067); 2: 4-arg (:828-897)."""
    src, tgt = _pts(src), _pts(tgt)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    t = np.ascontiguousarray(trunc, dtype=np.float64)
    m = ((C.c_double * 5) * 5)()
    n = C.c_int64(0)
    with _Workdir() as wd:
        _check(lib().ref_diff_reg_result(variant, _dp(src), len(src), _dp(tgt), len(tgt), _ip(pairs), len(pairs), _dp(t), wd.encode(),
                                         C.byref(m), C.byref(n)))
    a = np.array([[m[i][k] for k in range(5)] for i in range(5)])
    return {ROWS[i]: a[i] for i in range(n.value)}


def chamfer(a, b) -> float:
    a, b = _pts(a), _pts(b)
    cd = C.c_double(0)
    with _Workdir() as wd:
        _check(lib().ref_chamfer(_dp(a), len(a), _dp(b), len(b), wd.encode(), C.byref(cd)))
    return cd.value


def mme(variant: int, xyz, radius: float):
    """variant 0: ComputeMeanMapEntropy (serial, k>=5); 1: ...UsingNormal (OpenMP, k>=10); 2: ...UsingNormalTBB (k>=10).
    -> (mean, entropies[n], valid[n])"""
    xyz = _pts(xyz)
    ent = np.zeros(len(xyz))
    valid = np.zeros(len(xyz), np.uint8)
    mean = C.c_double(0)
    with _Workdir() as wd:
        _check(lib().ref_mme(variant, _dp(xyz), len(xyz), radius, wd.encode(), _dp(ent), valid.ctypes.data_as(C.POINTER(C.c_uint8)),
                             C.byref(mean)))
    return mean.value, ent, valid.astype(bool)


def mme_raw(variant: int, xyz, radius: float):
    """mme() plus the RAW bits of valid_entropy_points as the reference's loop left them: in the parallel variants they are written
    from several threads into a std::vector<bool> (map_eval.cpp:1586, :1694) — a lost update clears a flag whose entropy was stored.
    -> (mean, entropies[n], valid[n] (race-free reconstruction), valid_raw[n])"""
    xyz = _pts(xyz)
    ent = np.zeros(len(xyz))
    valid = np.zeros(len(xyz), np.uint8)
    raw = np.zeros(len(xyz), np.uint8)
    mean = C.c_double(0)
    u8 = C.POINTER(C.c_uint8)
    with _Workdir() as wd:
        _check(lib().ref_mme_raw(variant, _dp(xyz), len(xyz), radius, wd.encode(), _dp(ent), valid.ctypes.data_as(u8),
                                 raw.ctypes.data_as(u8), C.byref(mean)))
    return mean.value, ent, valid.astype(bool), raw.astype(bool)


def compute_entropy(cov) -> float:
    cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(9)
    return lib().ref_compute_entropy(_dp(cov))


def vmd(est, gt, voxel: float, workdir=None) -> dict:
    """calculateVMD (:240-390): AWD, SCS + the two files it writes."""
    est, gt = _pts(est), _pts(gt)
    a, s = C.c_double(0), C.c_double(0)
    with _Workdir(workdir) as wd:
        _check(lib().ref_vmd(_dp(est), len(est), _dp(gt), len(gt), voxel, wd.encode(), C.byref(a), C.byref(s)))
        files = _read_outputs(wd)
    return {"vmd": a.value, "scs": s.value, "files": files}


def _read_outputs(wd: str) -> dict:
    out = {}
    d = os.path.join(wd, "map_results")
    for name in ("voxel_errors.txt", "voxel_wasserstein_cdf.txt"):
        p = os.path.join(d, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            out[name] = np.loadtxt(p, ndmin=2)
        elif os.path.exists(p):
            out[name] = np.zeros((0, 27 if "errors" in name else 2))
    p = os.path.join(d, "map_results.txt")
    if os.path.exists(p):
        out["map_results.txt"] = open(p).read()
    return out


class VoxelMap:
    """VoxelCalculator::buildVoxelMap(cloud) (voxel_calculator.cpp:21-56), exported in ascending key order."""

    def __init__(self, xyz, voxel: float):
        xyz = _pts(xyz)
        self.voxel = voxel
        self.h = lib().ref_voxel_build(_dp(xyz), len(xyz), voxel)
        if not self.h:
            raise RuntimeError("reference: " + lib().ref_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_voxel_free(self.h)
            self.h = None

    def __len__(self):
        return lib().ref_voxel_count(self.h)

    def export(self) -> dict:
        v = len(self)
        keys, npts, active = np.zeros((v, 3), np.int32), np.zeros(v, np.int32), np.zeros(v, np.int32)
        mu, sigma, ent, en = np.zeros((v, 3)), np.zeros((v, 3, 3)), np.zeros(v), np.zeros(v)
        lib().ref_voxel_export(self.h, _ip(keys), _ip(npts), _dp(mu), _dp(sigma), _dp(ent), _dp(en), _ip(active))
        return {"keys": keys, "npts": npts, "mu": mu, "sigma": sigma, "entropy": ent, "energy": en, "active": active}

    def update_from(self, gt: "VoxelMap"):
        """est.updateVoxelMap(gt.getVoxelMap()) (:142-172) -> (active, old, new)"""
        c = (C.c_int64 * 3)()
        lib().ref_voxel_update(self.h, gt.h, c)
        return tuple(c)


def w2_gaussian(mu1, sigma1, n1, mu2, sigma2, n2) -> float:
    a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (mu1, sigma1, mu2, sigma2)]
    return lib().ref_w2_gaussian(_dp(a[0]), _dp(a[1]), int(n1), _dp(a[2]), _dp(a[3]), int(n2))


def voxel_index(p, voxel: float) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.float64).reshape(3)
    out = np.zeros(3, np.int32)
    lib().ref_voxel_index(_dp(p), voxel, _ip(out))
    return out


def neighbor_indices(index, radius: int) -> np.ndarray:
    idx = np.ascontiguousarray(index, dtype=np.int32).reshape(3)
    n = lib().ref_neighbor_indices(_ip(idx), radius, None)
    out = np.zeros((n, 3), np.int32)
    lib().ref_neighbor_indices(_ip(idx), radius, _ip(out))
    return out


def kdtree_nn1(ref, query):
    ref, query = _pts(ref), _pts(query)
    idx, d2 = np.zeros(len(query), np.int32), np.zeros(len(query))
    lib().ref_kdtree_nn1(_dp(ref), len(ref), _dp(query), len(query), _ip(idx), _dp(d2))
    return idx, d2


def kdtree_radius_count(ref, query, r: float):
    ref, query = _pts(ref), _pts(query)
    cnt = np.zeros(len(query), np.int32)
    lib().ref_kdtree_radius_count(_dp(ref), len(ref), _dp(query), len(query), r, _ip(cnt))
    return cnt
