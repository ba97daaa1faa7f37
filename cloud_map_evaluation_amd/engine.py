"""Python face of the C ABI: numpy / torch-tensor in, numpy / dataclasses out.

`Engine` owns one me_ctx (one GPU).  Method names follow the reference's MapEval members they stand in for
(map_eval/src/map_eval.h:196-312): computeMME, calculateMetricsWithInitialMatrix, computeChamferDistance,
calculateVMD.  The C++ host (cloud_map_evaluation_amd/host/) is the drop-in for the reference executable; this
module is what tests/ and bench.py drive.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib
from ._lib import ME_GATE_LE_UNSQUARED, ME_GATE_LT_SQUARED, ME_SLOT_EST, ME_SLOT_GT  # noqa: F401


class MapEvalError(RuntimeError):
    pass


@dataclass
class Param:
    """Hot-path fields of the reference's Param (map_eval.h:60-116), same names and defaults."""
    icp_max_distance_: float = 2.5
    nn_radius_: float = 0.2
    trunc_dist_: tuple = (0.2, 0.1, 0.08, 0.05, 0.01)  # accuracy_level (config.yaml)
    initial_matrix_: np.ndarray = field(default_factory=lambda: np.eye(4))
    vmd_voxel_size_: float = 3.0
    evaluate_mme_: bool = True
    evaluate_gt_mme_: bool = True
    evaluate_using_initial_: bool = True
    use_tbb_mme: bool = True  # accepted for compatibility; the GPU path has one implementation


@dataclass
class RegStats:
    """One getDiffRegResultWithCorrespondence result block (map_eval.cpp:1140-1144)."""
    n_src: int
    n_corr: int
    mean: np.ndarray
    rmse: np.ndarray
    fitness: np.ndarray
    sigma: np.ndarray
    number: np.ndarray
    mean_nn_dist: float

    @staticmethod
    def from_c(o: _lib.NNStatsOut) -> "RegStats":
        f = lambda x: np.array(list(x), dtype=np.float64)
        return RegStats(o.n_src, o.n_corr, f(o.mean), f(o.rmse), f(o.fitness), f(o.sigma), f(o.number), o.mean_nn_dist)


def _addr(a) -> int:
    """Raw address of a numpy array (host) or a torch tensor (host or device)."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch.Tensor


class Engine:
    def __init__(self, device: int = 0, borrow_device_input: bool = False, morton_order: bool = False):
        """borrow_device_input: ME_FLAG_BORROW_DEVICE_INPUT — cuda tensors uploaded without a transform are read where they lie
        (no copy); the Engine keeps a reference to them until the slot's next upload, the caller must not modify them meanwhile.
        morton_order: ME_FLAG_MORTON_ORDER — Z curve instead of the Hilbert curve (tests / measurements)."""
        self._L = _lib.load()
        self._ctx = self._L.me_create(int(device), (1 if borrow_device_input else 0) | (2 if morton_order else 0))
        self._held = {}  # slot -> the array / tensor of its last upload (borrowed device inputs must outlive their use)
        if not self._ctx:
            raise MapEvalError(self._L.me_last_error(None).decode())
        self.device = device

    def twin(self) -> "Engine":
        """A second lane on the same clouds (me_twin): its own stream and scratch, so that a second host thread can run
        independent work concurrently (ctypes calls release the GIL).  Owned by this engine."""
        if getattr(self, "_twin", None) is None:
            t = Engine.__new__(Engine)
            t._L = self._L
            t._ctx = self._L.me_twin(self._ctx)
            if not t._ctx:
                raise MapEvalError(self._L.me_last_error(self._ctx).decode())
            t.device = self.device
            t._held = self._held
            t._owned = False
            t._twin = None
            self._twin = t
        return self._twin

    def close(self):
        if getattr(self, "_ctx", None):
            if getattr(self, "_owned", True):
                self._L.me_destroy(self._ctx)
                if getattr(self, "_twin", None) is not None:
                    self._twin._ctx = None  # freed with the primary context
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc: int):
        if rc != 0:
            raise MapEvalError(f"[{rc}] " + self._L.me_last_error(self._ctx).decode())

    # ---- clouds ----
    def set_shard(self, rank: int, world: int):
        self._ck(self._L.me_set_shard(self._ctx, rank, world))

    def upload(self, slot: int, xyz, T=None, cell_size: float = 0.0):
        """xyz: (N,3) float64 numpy array (host) or torch tensor (CPU or cuda, contiguous)."""
        Tm = None if T is None else np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        on_device = False
        if isinstance(xyz, np.ndarray):
            xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        else:  # torch tensor
            import torch

            if xyz.dtype != torch.float64 or not xyz.is_contiguous():
                xyz = xyz.to(torch.float64).contiguous()
            on_device = xyz.is_cuda
            if on_device:
                torch.cuda.current_stream(xyz.device).synchronize()  # producer stream -> library stream hand-over
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("expected an (N,3) array")
        fn = self._L.me_upload_cloud_device if on_device else self._L.me_upload_cloud
        self._ck(fn(self._ctx, slot, _addr(xyz), int(xyz.shape[0]), _addr(Tm), float(cell_size)))
        self._held[slot] = xyz

    def upload_slab(self, slot: int, xyz, cell_size: float = 0.0):
        """upload() for a cuda tensor that already IS this rank's slab + halo (what the halo exchange delivered): no filter."""
        import torch

        if not xyz.is_cuda:
            return self.upload(slot, xyz, cell_size=cell_size)  # (host tensor: the general path, with its filter)
        if xyz.dtype != torch.float64 or not xyz.is_contiguous():
            xyz = xyz.to(torch.float64).contiguous()
        torch.cuda.current_stream(xyz.device).synchronize()
        self._ck(self._L.me_upload_slab_device(self._ctx, slot, xyz.data_ptr(), int(xyz.shape[0]), float(cell_size)))
        self._held[slot] = xyz

    def voxel_downsample(self, slot: int, voxel_size: float) -> int:
        """open3d VoxelDownSample (map_eval.cpp:38-39) on the uploaded cloud, in place; returns the new point count."""
        n = C.c_int64(0)
        self._ck(self._L.me_voxel_downsample(self._ctx, slot, float(voxel_size), C.byref(n)))
        return n.value

    def transform_cloud(self, slot: int, T):
        """*cloud = cloud->Transform(T) (map_eval.cpp:1206) on the device."""
        Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        self._ck(self._L.me_transform_cloud(self._ctx, slot, _addr(Tm)))

    def size(self, slot: int) -> int:
        return int(self._L.me_cloud_size(self._ctx, slot))

    def download(self, slot: int) -> np.ndarray:
        out = np.empty((self.size(slot), 3), np.float64)
        self._ck(self._L.me_download_cloud(self._ctx, slot, _addr(out)))
        return out

    # ---- 1-NN + AC/COM/CD ----
    def nn1(self, query_slot: int, ref_slot: int, fetch: bool = True):
        n = self.size(query_slot)
        if not fetch:
            self._ck(self._L.me_nn1(self._ctx, query_slot, ref_slot, 0, 0))
            return None, None
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float64)
        self._ck(self._L.me_nn1(self._ctx, query_slot, ref_slot, _addr(idx), _addr(d2)))
        return idx, d2

    def nn_stats(self, query_slot: int, gate: float, gate_mode: int, trunc) -> RegStats:
        tr = np.ascontiguousarray(trunc, dtype=np.float64)
        out = _lib.NNStatsOut()
        self._ck(self._L.me_nn_stats(self._ctx, query_slot, float(gate), int(gate_mode), _addr(tr), C.byref(out)))
        return RegStats.from_c(out)

    def nn_partial_sums(self, query_slot: int, gate: float, gate_mode: int, trunc) -> _lib.NNPartial:
        tr = np.ascontiguousarray(trunc, dtype=np.float64)
        out = _lib.NNPartial()
        self._ck(self._L.me_nn_partial_sums(self._ctx, query_slot, float(gate), int(gate_mode), _addr(tr), C.byref(out)))
        return out

    def nn_sigma_sums(self, query_slot: int, gate: float, gate_mode: int, mean) -> np.ndarray:
        m = np.ascontiguousarray(mean, dtype=np.float64)
        out = np.zeros(5, np.float64)
        self._ck(self._L.me_nn_sigma_sums(self._ctx, query_slot, float(gate), int(gate_mode), _addr(m), _addr(out)))
        return out

    def nn_finalize(self, total: _lib.NNPartial, sigma_num, n_src_total: int) -> RegStats:
        s = np.ascontiguousarray(sigma_num, dtype=np.float64)
        out = _lib.NNStatsOut()
        self._L.me_nn_finalize(C.byref(total), _addr(s), int(n_src_total), C.byref(out))
        return RegStats.from_c(out)

    def icp_p2p_sums(self, query_slot: int, max_distance: float) -> _lib.IcpSums:
        """Sums of the point-to-point ICP step over the correspondences (d2 < max^2) of the last nn1(query_slot, ...)."""
        out = _lib.IcpSums()
        self._ck(self._L.me_icp_p2p_sums(self._ctx, query_slot, float(max_distance), C.byref(out)))
        return out

    def renderDistanceOnPointCloud(self, query_slot: int, dis: float, gate: float = -1.0, gate_mode: int = 0):
        """map_eval.cpp:586-607 for the last nn1(query_slot, ...) -> (rgb (N,3), inlier (N,) bool)."""
        n = self.size(query_slot)
        rgb = np.empty((n, 3), np.float64)
        inl = np.empty(n, np.uint8)
        self._ck(self._L.me_render_distance(self._ctx, query_slot, float(dis), float(gate), int(gate_mode), _addr(rgb), _addr(inl)))
        return rgb, inl.astype(bool)

    def ColorPointCloudByMME(self, slot: int):
        """map_eval.cpp:686-735 for the last mme(slot) -> (xyz_valid (M,3), rgb (M,3), min_abs, max_abs)."""
        m, mn, mx = C.c_int64(0), C.c_double(), C.c_double()
        self._ck(self._L.me_render_entropy(self._ctx, slot, 0, 0, 0, C.byref(m), C.byref(mn), C.byref(mx)))
        xyz = np.empty((m.value, 3), np.float64)
        rgb = np.empty((m.value, 3), np.float64)
        if m.value:
            self._ck(self._L.me_render_entropy(self._ctx, slot, _addr(xyz), _addr(rgb), m.value, C.byref(m), C.byref(mn), C.byref(mx)))
        return xyz, rgb, mn.value, mx.value

    def set_normals(self, slot: int, normals) -> None:
        """Normals that came with the cloud (N,3), caller's point order."""
        nrm = np.ascontiguousarray(normals, dtype=np.float64)
        if nrm.shape != (self.size(slot), 3):
            raise ValueError("normals must be (N,3) for the N points of the slot")
        self._ck(self._L.me_set_normals(self._ctx, slot, _addr(nrm)))

    def get_normals(self, slot: int) -> np.ndarray:
        out = np.empty((self.size(slot), 3), np.float64)
        self._ck(self._L.me_get_normals(self._ctx, slot, _addr(out)))
        return out

    def estimate_normals(self, slot: int, knn: int = 20, fetch: bool = True, with_neighbours: bool = False):
        """open3d EstimateNormals(KDTreeSearchParamKNN(knn)) -> normals (N,3) [, knn_idx (N,knn), knn_d2 (N,knn)]."""
        n = self.size(slot)
        nrm = np.empty((n, 3), np.float64) if fetch else None
        idx = np.empty((n, knn), np.int32) if with_neighbours else None
        d2 = np.empty((n, knn), np.float64) if with_neighbours else None
        self._ck(self._L.me_estimate_normals(self._ctx, slot, int(knn), _addr(nrm) if fetch else 0,
                                             _addr(idx) if with_neighbours else 0, _addr(d2) if with_neighbours else 0))
        return (nrm, idx, d2) if with_neighbours else nrm

    def gicp_covariances(self, slot: int, epsilon: float = 1e-3, fetch: bool = False):
        """open3d InitializePointCloudForGeneralizedICP -> (N,3,3) when fetch."""
        out = np.empty((self.size(slot), 9), np.float64) if fetch else None
        self._ck(self._L.me_gicp_covariances(self._ctx, slot, float(epsilon), _addr(out) if fetch else 0))
        return out.reshape(-1, 3, 3) if fetch else None

    def get_covariances(self, slot: int) -> np.ndarray:
        out = np.empty((self.size(slot), 9), np.float64)
        self._ck(self._L.me_get_covariances(self._ctx, slot, _addr(out)))
        return out.reshape(-1, 3, 3)

    def icp_lsq_sums(self, query_slot: int, mode: int, max_distance: float) -> _lib.IcpLsq:
        """J^T J, J^T r of one point-to-plane (mode 1) / generalized (mode 2) step over the last nn1(query_slot, ...)."""
        out = _lib.IcpLsq()
        self._ck(self._L.me_icp_lsq_sums(self._ctx, query_slot, int(mode), float(max_distance), C.byref(out)))
        return out

    def performICPRegistration(self, max_distance: float, method: int = 0, **criteria):
        """map_eval.cpp:1366-1394: registration_methods 0 point-to-point, 1 point-to-plane, 2 generalized ICP (see icp.py)."""
        from . import icp
        if method == 0:
            return icp.icp_point_to_point(self, max_distance, **criteria)
        if method == 1:
            return icp.icp_point_to_plane(self, max_distance, **criteria)
        if method == 2:
            return icp.icp_generalized(self, max_distance, **criteria)
        raise MapEvalError("Invalid registration type specified")  # (:1385-1387)

    def computeChamferDistance(self) -> float:
        """map_eval.cpp:1398-1431 on the uploaded pair."""
        cd = C.c_double()
        self._ck(self._L.me_chamfer(self._ctx, C.byref(cd)))
        return cd.value

    def calculateMetricsWithInitialMatrix(self, p: Param):
        """map_eval.cpp:1204-1260 (clouds already uploaded, est with initial_matrix_) -> (est_gt, gt_est, cd_vec)."""
        self.nn1(ME_SLOT_EST, ME_SLOT_GT, fetch=False)
        est_gt = self.nn_stats(ME_SLOT_EST, p.icp_max_distance_, ME_GATE_LE_UNSQUARED, p.trunc_dist_)
        self.nn1(ME_SLOT_GT, ME_SLOT_EST, fetch=False)
        gt_est = self.nn_stats(ME_SLOT_GT, p.icp_max_distance_, ME_GATE_LE_UNSQUARED, p.trunc_dist_)
        return est_gt, gt_est, est_gt.rmse + gt_est.rmse  # cd_vec (:1245)

    # ---- MME ----
    def mme(self, slot: int, radius: float, min_k: int, per_point: bool = True):
        """-> (mean, entropies[N] | None, valid[N] | None, n_valid, sum_H)."""
        n = self.size(slot)
        ent = np.zeros(n, np.float64) if per_point else None
        val = np.zeros(n, np.uint8) if per_point else None
        s = C.c_double()
        nv = C.c_int64()
        self._ck(self._L.me_mme(self._ctx, slot, float(radius), int(min_k), _addr(ent), _addr(val), C.byref(s), C.byref(nv)))
        mean = s.value / nv.value if nv.value > 0 else 0.0
        return mean, ent, val, nv.value, s.value

    def computeMME(self, p: Param):
        """map_eval.cpp:149-189 -> (mme_est, mme_gt)."""
        mme_est = self.mme(ME_SLOT_EST, p.nn_radius_, 10, per_point=False)[0]
        mme_gt = self.mme(ME_SLOT_GT, p.nn_radius_, 5, per_point=False)[0] if p.evaluate_gt_mme_ else 0.0
        return mme_est, mme_gt

    # ---- voxels ----
    def voxel_build(self, slot: int, voxel_size: float) -> int:
        """Builds (and caches on the cloud) the voxel-Gaussian table without exporting it; returns the voxel count."""
        nv = C.c_int64(0)
        self._ck(self._L.me_voxel_gaussians(self._ctx, slot, float(voxel_size), 0, 0, 0, 0, 0, C.byref(nv)))
        return nv.value

    def voxel_gaussians(self, slot: int, voxel_size: float):
        nv = C.c_int64(0)
        self._ck(self._L.me_voxel_gaussians(self._ctx, slot, float(voxel_size), 0, 0, 0, 0, 0, C.byref(nv)))
        v = nv.value
        keys = np.empty((v, 3), np.int32)
        n = np.empty(v, np.int32)
        mu = np.empty((v, 3), np.float64)
        sg = np.empty((v, 9), np.float64)
        en = np.empty(v, np.float64)
        nv = C.c_int64(v)
        self._ck(self._L.me_voxel_gaussians(self._ctx, slot, float(voxel_size), _addr(keys), _addr(n), _addr(mu), _addr(sg),
                                            _addr(en), C.byref(nv)))
        return keys, n, mu, sg.reshape(v, 3, 3), en

    def calculateVMD(self, voxel_size: float, min_pts: int = 100, scs_radius: int = 5, rows: bool = True):
        """map_eval.cpp:240-390 -> dict(awd, scs, rows, w_sorted, counts)."""
        n = C.c_int64(0)
        awd, scs = C.c_double(), C.c_double()
        counts = np.zeros(3, np.int64)
        self._ck(self._L.me_awd_scs(self._ctx, float(voxel_size), min_pts, scs_radius, 0, 0, C.byref(n), C.byref(awd),
                                    C.byref(scs), _addr(counts)))
        res = dict(awd=awd.value, scs=scs.value, counts=tuple(int(c) for c in counts), n_rows=n.value)
        if rows and n.value > 0:
            r = np.empty((n.value, 27), np.float64)
            ws = np.empty(n.value, np.float64)
            cap = C.c_int64(n.value)
            self._ck(self._L.me_awd_scs(self._ctx, float(voxel_size), min_pts, scs_radius, _addr(r), _addr(ws), C.byref(cap),
                                        C.byref(awd), C.byref(scs), _addr(counts)))
            res.update(rows=r, w_sorted=ws)
        elif rows:
            res.update(rows=np.empty((0, 27)), w_sorted=np.empty(0))
        return res

    def w2_batch(self, mu1, sigma1, n1, mu2, sigma2, n2) -> np.ndarray:
        """Batched computeWassersteinDistanceGaussian(voxel1, voxel2) (voxel_calculator.cpp:115-140)."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (mu1, sigma1, mu2, sigma2)]
        n1 = np.ascontiguousarray(n1, dtype=np.int32)
        n2 = np.ascontiguousarray(n2, dtype=np.int32)
        cnt = n1.shape[0]
        w = np.empty(cnt, np.float64)
        self._ck(self._L.me_w2_batch(self._ctx, _addr(a[0]), _addr(a[1]), _addr(n1), _addr(a[2]), _addr(a[3]), _addr(n2), cnt,
                                     _addr(w)))
        return w

    def scs_table(self, keys, w, radius: int = 5) -> float:
        """SCS (map_eval.cpp:347-389) of a sparse W table."""
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = C.c_double()
        self._ck(self._L.me_scs_table(self._ctx, _addr(keys), _addr(w), w.shape[0], radius, C.byref(out)))
        return out.value

    # ---- multi-GPU spatial slab mode (include/mapeval_hip.h: me_set_slab ...) ----
    def set_slab(self, axis: int, lo: float = 0.0, hi: float = 0.0, halo: float = 0.0):
        """Keep lo-halo <= p[axis] < hi+halo at the next uploads; lo <= p[axis] < hi are the points this rank owns."""
        self._ck(self._L.me_set_slab(self._ctx, int(axis), float(lo), float(hi), float(halo)))

    def nn_unresolved_count(self, query_slot: int) -> int:
        n = C.c_int64(0)
        self._ck(self._L.me_nn_unresolved(self._ctx, query_slot, 0, 0, 0, C.byref(n)))
        return n.value

    def nn_unresolved(self, query_slot: int, with_d2: bool = False):
        """(count, 3) float64 cuda tensor: owned queries whose 1-NN may live on another rank; with_d2: (count, 4), the
        fourth column = their current best squared distance (the bound the other ranks have to beat)."""
        import torch

        cnt = self.nn_unresolved_count(query_slot)
        dev = torch.device("cuda", self.device)
        out = torch.empty((cnt, 3), dtype=torch.float64, device=dev)
        d2 = torch.empty(cnt, dtype=torch.float64, device=dev) if with_d2 else None
        if cnt:
            n = C.c_int64(0)
            self._ck(self._L.me_nn_unresolved(self._ctx, query_slot, out.data_ptr(), d2.data_ptr() if with_d2 else 0, cnt,
                                              C.byref(n)))
        return torch.cat([out, d2[:, None]], 1) if with_d2 else out

    def nn_points(self, ref_slot: int, xyz, bound=None, covered=None, axis: int = 0):
        """Exact squared distance of arbitrary points (cuda tensor (m,3) float64) to this rank's part of ref_slot;
        bound (m,): upper bounds -> min(bound, nearest here), far ranks prune at once (me_nn_points_bounded);
        covered (m,2) + axis: the band [lo, hi) of `axis` each query's owner has searched already (me_nn_points_covered)."""
        import torch

        xyz = xyz.to(torch.device("cuda", self.device), torch.float64).contiguous()
        cov = None
        if bound is None:
            d2 = torch.empty(xyz.shape[0], dtype=torch.float64, device=xyz.device)
            fn = self._L.me_nn_points
        else:
            d2 = bound.to(xyz.device, torch.float64).clone().contiguous()
            fn = self._L.me_nn_points_bounded
            if covered is not None:
                cov = covered.to(xyz.device, torch.float64).contiguous()
        torch.cuda.current_stream(xyz.device).synchronize()
        if cov is not None:
            self._ck(self._L.me_nn_points_covered(self._ctx, ref_slot, xyz.data_ptr(), int(xyz.shape[0]), d2.data_ptr(), int(axis), cov.data_ptr()))
        else:
            self._ck(fn(self._ctx, ref_slot, xyz.data_ptr(), int(xyz.shape[0]), d2.data_ptr()))
        return d2

    def nn_fetch(self, query_slot: int):
        """-> (idx, d2) of the last nn1(query_slot, ..) as it stands now (slab mode: after nn_patch; halo points d2 = -1)."""
        n = self.size(query_slot)
        idx, d2 = np.empty(n, np.int32), np.empty(n, np.float64)
        self._ck(self._L.me_nn_fetch(self._ctx, query_slot, _addr(idx), _addr(d2)))
        return idx, d2

    def set_mme_result(self, slot: int, entropies, valid):
        """Per-point MME results computed elsewhere (distributed run) -> ColorPointCloudByMME works on this context."""
        e = np.ascontiguousarray(entropies, dtype=np.float64)
        v = np.ascontiguousarray(valid, dtype=np.uint8)
        self._ck(self._L.me_set_mme_result(self._ctx, slot, _addr(e), _addr(v)))

    def set_nn_result(self, query_slot: int, ref_slot: int, d2):
        d = np.ascontiguousarray(d2, dtype=np.float64)
        self._ck(self._L.me_set_nn_result(self._ctx, query_slot, ref_slot, _addr(d)))

    def slab_points(self, slot: int):
        """Slab mode: (orig_index[n], owned[n]) of the points this context holds, in the order of its per-point outputs."""
        n = self.size(slot)
        orig, owned = np.empty(n, np.int64), np.empty(n, np.uint8)
        cnt = C.c_int64(0)
        self._ck(self._L.me_slab_points(self._ctx, slot, _addr(orig), _addr(owned), n, C.byref(cnt)))
        return orig, owned.astype(bool)

    # ---- the cross-rank 1-NN step on one fixed-capacity message (me_nn_cross_*) ----
    def nn_cross_message(self, cap: int, n_loc_est: int, n_loc_gt: int):
        """-> (message (1 + 2 cap, 4) cuda tensor, [open queries map -> gt, gt -> map])."""
        import torch

        msg = torch.empty((1 + 2 * cap, 4), dtype=torch.float64, device=torch.device("cuda", self.device))
        counts = np.zeros(2, np.int64)
        self._ck(self._L.me_nn_cross_message(self._ctx, msg.data_ptr(), int(cap), int(n_loc_est), int(n_loc_gt), _addr(counts)))
        return msg, [int(counts[0]), int(counts[1])]

    def nn_cross_answer(self, gathered, cap: int, own_rank: int, dir_mask: int, axis: int, cuts, halo: float):
        """gathered (world, 1 + 2 cap, 4) cuda tensor -> d2 (world, 1 + 2 cap): the block the ranks min-reduce."""
        import torch

        g = gathered.to(torch.device("cuda", self.device), torch.float64).contiguous()
        world = int(g.shape[0])
        d2 = torch.zeros((world, 1 + 2 * cap), dtype=torch.float64, device=g.device)
        c = np.ascontiguousarray(cuts, dtype=np.float64)
        torch.cuda.current_stream(g.device).synchronize()
        self._ck(self._L.me_nn_cross_answer(self._ctx, g.data_ptr(), world, int(cap), int(own_rank), int(dir_mask), int(axis), _addr(c), float(halo),
                                            d2.data_ptr()))
        return d2

    def nn_cross_patch(self, d2_reduced, cap: int, own_rank: int):
        import torch

        d = d2_reduced.to(torch.device("cuda", self.device), torch.float64).contiguous()
        torch.cuda.current_stream(d.device).synchronize()
        self._ck(self._L.me_nn_cross_patch(self._ctx, d.data_ptr(), int(cap), int(own_rank)))

    def nn_patch(self, query_slot: int, d2):
        import torch

        d2 = d2.to(torch.device("cuda", self.device), torch.float64).contiguous()
        torch.cuda.current_stream(d2.device).synchronize()
        self._ck(self._L.me_nn_patch(self._ctx, query_slot, d2.data_ptr(), int(d2.shape[0])))

    def voxel_partials(self, slot: int, voxel_size: float):
        """Owned-point voxel partials: keys[V,3] int32, n[V] int32, mu[V,3], raw M2[V,3,3]."""
        nv = C.c_int64(0)
        self._ck(self._L.me_voxel_partials(self._ctx, slot, float(voxel_size), 0, 0, 0, 0, C.byref(nv)))
        v = nv.value
        keys = np.empty((v, 3), np.int32)
        n = np.empty(v, np.int32)
        mu = np.empty((v, 3), np.float64)
        m2 = np.empty((v, 9), np.float64)
        if v:
            nv = C.c_int64(v)
            self._ck(self._L.me_voxel_partials(self._ctx, slot, float(voxel_size), _addr(keys), _addr(n), _addr(mu), _addr(m2),
                                               C.byref(nv)))
        return keys, n, mu, m2.reshape(v, 3, 3)

    # ---- multi-GPU with distributed input (include/mapeval_hip.h: me_halo_pack_device ...) ----
    def transform_points(self, xyz, T):
        """Open3D Transform (map_eval.cpp:1206) on a cuda tensor (n,3) float64, in place; returns it."""
        import torch

        Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        assert xyz.is_cuda and xyz.dtype == torch.float64 and xyz.is_contiguous()
        torch.cuda.current_stream(xyz.device).synchronize()
        self._ck(self._L.me_transform_points_device(self._ctx, xyz.data_ptr(), int(xyz.shape[0]), _addr(Tm)))
        return xyz

    def halo_pack(self, xyz, axis: int, cuts, halo: float):
        """Send side of the halo exchange: xyz (n,3) cuda float64 -> (packed (m,3) cuda tensor, destination-major; counts
        list[world]).  Rank k receives every point with cuts[k] - halo <= p[axis] < cuts[k+1] + halo."""
        import torch

        world = len(cuts) - 1
        c = np.ascontiguousarray(cuts, dtype=np.float64)
        counts = np.zeros(world, np.int64)
        assert xyz.is_cuda and xyz.dtype == torch.float64 and xyz.is_contiguous()
        torch.cuda.current_stream(xyz.device).synchronize()
        n = int(xyz.shape[0])
        # one call: count + scatter into a buffer sized for the usual case (every point to its owner, a thin halo to the
        # neighbours); the rare overflow (halo wider than the slabs) comes back as ME_ERR_CAPACITY with the exact counts
        cap = n + n // 2 + 4096
        out = torch.empty((cap, 3), dtype=torch.float64, device=xyz.device)
        rc = self._L.me_halo_pack_device(self._ctx, xyz.data_ptr(), n, int(axis), _addr(c), world, float(halo), out.data_ptr(), cap,
                                         _addr(counts))
        total = int(counts.sum())
        if rc == _lib.ME_ERR_CAPACITY:
            out = torch.empty((total, 3), dtype=torch.float64, device=xyz.device)
            rc = self._L.me_halo_pack_device(self._ctx, xyz.data_ptr(), n, int(axis), _addr(c), world, float(halo), out.data_ptr(),
                                             total, _addr(counts))
        self._ck(rc)
        out = out[:total]
        return out, [int(x) for x in counts]

    def lattice_messages(self, parts, e0: int):
        """The rows of this rank's plan-gather message for its 1 or 2 pieces (cuda (n,3) float64 tensors): int64 (len(parts),
        8 + 3 ME_LATTICE_BINS) on the device, header included — me_lattice_messages_device."""
        import torch

        dev = torch.device("cuda", self.device)
        for p in parts:
            assert p.is_cuda and p.dtype == torch.float64 and p.is_contiguous()
        torch.cuda.current_stream(dev).synchronize()
        msg = torch.empty((len(parts), 8 + 3 * _lib.ME_LATTICE_BINS), dtype=torch.int64, device=dev)
        a = parts[0]
        b = parts[1] if len(parts) > 1 else parts[0]
        self._ck(self._L.me_lattice_messages_device(self._ctx, a.data_ptr(), int(a.shape[0]), b.data_ptr(), int(b.shape[0]) if len(parts) > 1 else 0,
                                                    len(parts), int(e0), msg.data_ptr()))
        return msg

    def lattice_plan_raw(self, allm, halo: float, e0: int):
        """The plan of the lean exchange from the gathered messages (world, clouds, 8 + 3 ME_LATTICE_BINS) int64 cuda tensor:
        me_lattice_plan_device's output vector as numpy int64 (dist.lattice_plan unpacks it)."""
        import torch

        g = allm.to(torch.device("cuda", self.device), torch.int64).contiguous()
        world, clouds = int(g.shape[0]), int(g.shape[1])
        out = np.zeros(4 + clouds + world - 1 + world * clouds * world, np.int64)
        torch.cuda.current_stream(g.device).synchronize()
        self._ck(self._L.me_lattice_plan_device(self._ctx, g.data_ptr(), world, clouds, float(halo), int(e0), out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def set_voxel_hint(self, voxel_size: float):
        """Index builds from now on also emit the voxel run records for this voxel size (0: off): me_set_voxel_hint."""
        self._ck(self._L.me_set_voxel_hint(self._ctx, float(voxel_size)))

    def lattice_histograms(self, xyz, e0: int):
        """Marginal histograms of a raw cuda (n,3) float64 buffer on the absolute lattice of bin width 2^(e0 + level):
        (level, origin_bin int64[3], neg_inf int64[3], hist (3, ME_LATTICE_BINS) cuda int32).  me_lattice_histograms_device."""
        import torch

        assert xyz.is_cuda and xyz.dtype == torch.float64 and xyz.is_contiguous()
        torch.cuda.current_stream(xyz.device).synchronize()
        hist = torch.empty((3, _lib.ME_LATTICE_BINS), dtype=torch.int32, device=xyz.device)
        level = C.c_int32(0)
        origin = (C.c_int64 * 3)()
        ninf = (C.c_int64 * 3)()
        self._ck(self._L.me_lattice_histograms_device(self._ctx, xyz.data_ptr(), int(xyz.shape[0]), int(e0), C.byref(level), origin, ninf,
                                                      hist.data_ptr()))
        return int(level.value), np.array(list(origin), dtype=np.int64), np.array(list(ninf), dtype=np.int64), hist

    def voxel_partial_rows(self, slot: int, voxel_size: float):
        """This rank's voxel partials as a (V,16) cuda tensor [kx,ky,kz,n,mu(3),M2(9)] (no host copy)."""
        import torch

        nv = C.c_int64(0)
        self._ck(self._L.me_voxel_partial_rows_device(self._ctx, slot, float(voxel_size), 0, 0, C.byref(nv)))
        rows = torch.empty((nv.value, 16), dtype=torch.float64, device=torch.device("cuda", self.device))
        if nv.value:
            self._ck(self._L.me_voxel_partial_rows_device(self._ctx, slot, float(voxel_size), rows.data_ptr(), nv.value, C.byref(nv)))
        return rows

    def voxel_merge(self, slot: int, voxel_size: float, rows):
        """Chan merge of the gathered partial rows of ALL ranks (cuda tensor (m,16); n == 0 rows are padding) into the slot's
        voxel table; calculateVMD then runs on the merged tables."""
        import torch

        rows = rows.to(torch.device("cuda", self.device), torch.float64).contiguous()
        torch.cuda.current_stream(rows.device).synchronize()
        self._ck(self._L.me_voxel_merge_device(self._ctx, slot, float(voxel_size), rows.data_ptr(), int(rows.shape[0])))

    # ---- whole suite ----
    def run_suite(self, p: Param, gate_mode: int = ME_GATE_LE_UNSQUARED) -> _lib.SuiteOut:
        sp = self._suite_params(p, gate_mode)
        out = _lib.SuiteOut()
        self._ck(self._L.me_run_suite(self._ctx, C.byref(sp), C.byref(out)))
        return out

    def _suite_params(self, p: Param, gate_mode: int) -> _lib.SuiteParams:
        sp = _lib.SuiteParams()
        sp.icp_max_distance = p.icp_max_distance_
        sp.gate_mode = gate_mode
        for k in range(5):
            sp.trunc[k] = p.trunc_dist_[k]
        sp.nn_radius = p.nn_radius_
        sp.vmd_voxel_size = p.vmd_voxel_size_
        sp.evaluate_mme = int(p.evaluate_mme_)
        sp.evaluate_gt_mme = int(p.evaluate_gt_mme_)
        sp.min_pts = 100
        sp.scs_radius = 5
        return sp

    def run_suite_from(self, est, gt, p: Param, overlap: bool = True, gate_mode: int = ME_GATE_LE_UNSQUARED,
                       pin_host_input: bool = False) -> _lib.SuiteOut:
        """me_run_suite_from: the whole pass of MapEval::process (map_eval.cpp:52-85) from the two raw clouds in ONE library call —
        uploads, index builds, MME x2 (the map as loaded), p.initial_matrix_ (:1206), both 1-NN directions + statistics, voxel
        Gaussians, AWD / CDF / SCS; overlap: the library's internal second lane (ME_SUITE_OVERLAP).  est / gt: (N,3) float64 numpy
        arrays or torch tensors (both host or both cuda); None, None: the clouds already uploaded."""
        flags = (_lib.ME_SUITE_OVERLAP if overlap else 0) | (_lib.ME_SUITE_PIN_HOST_INPUT if pin_host_input else 0)
        ne = ng = 0
        if est is not None:
            on_dev = []
            arrs = []
            for a in (est, gt):
                if isinstance(a, np.ndarray):
                    a = np.ascontiguousarray(a, dtype=np.float64)
                    on_dev.append(False)
                else:
                    import torch

                    if a.dtype != torch.float64 or not a.is_contiguous():
                        a = a.to(torch.float64).contiguous()
                    on_dev.append(bool(a.is_cuda))
                    if a.is_cuda:
                        torch.cuda.current_stream(a.device).synchronize()  # producer stream -> library stream hand-over
                if a.ndim != 2 or a.shape[1] != 3:
                    raise ValueError("expected (N,3) arrays")
                arrs.append(a)
            if on_dev[0] != on_dev[1]:
                raise ValueError("run_suite_from: both clouds in host memory, or both on the device")
            est, gt = arrs
            if on_dev[0]:
                flags |= _lib.ME_SUITE_DEVICE_INPUT
            ne, ng = int(est.shape[0]), int(gt.shape[0])
            self._held[ME_SLOT_EST], self._held[ME_SLOT_GT] = est, gt
        Tm = np.ascontiguousarray(p.initial_matrix_, dtype=np.float64).reshape(16)
        sp = self._suite_params(p, gate_mode)
        out = _lib.SuiteOut()
        self._ck(self._L.me_run_suite_from(self._ctx, _addr(est), ne, _addr(gt), ng, _addr(Tm), C.byref(sp), flags, C.byref(out)))
        return out

    @staticmethod
    def suite_dict(o: _lib.SuiteOut) -> dict:
        """A SuiteOut as the dict dist.suite_step returns (same keys, same numbers)."""
        def d(s):
            f = lambda x: np.array(list(x), dtype=np.float64)
            return dict(n_corr=int(s.n_corr), number=f(s.number), mean=f(s.mean), rmse=f(s.rmse), fitness=f(s.fitness),
                        sigma=f(s.sigma), mean_nn=float(s.mean_nn_dist))
        eg, ge = d(o.est_gt), d(o.gt_est)
        return dict(est_gt=eg, gt_est=ge, ac=eg["rmse"], com=eg["fitness"], cd=float(o.full_chamfer), mme_est=float(o.mme_est),
                    mme_gt=float(o.mme_gt), mme_valid=int(o.mme_est_valid), awd=float(o.awd), scs=float(o.scs), n_w=int(o.n_w_voxels),
                    n_est=int(o.est_gt.n_src), n_gt=int(o.gt_est.n_src), stage_ms=[float(x) for x in o.stage_ms])

    def mme_fetch(self, slot: int):
        """me_mme_fetch: (entropies[N], valid[N]) of the slot's last MME pass, cloud order."""
        n = self.size(slot)
        ent = np.zeros(n, np.float64)
        val = np.zeros(n, np.uint8)
        self._ck(self._L.me_mme_fetch(self._ctx, slot, _addr(ent), _addr(val)))
        return ent, val

    # ---- instrumentation ----
    def timers_enable(self, on: bool = True):
        self._ck(self._L.me_timers_enable(self._ctx, int(on)))

    def timers_reset(self):
        self._ck(self._L.me_timers_reset(self._ctx))

    def timer(self, name: str):
        ms, cnt = C.c_double(), C.c_int64()
        self._ck(self._L.me_timer_get(self._ctx, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value
