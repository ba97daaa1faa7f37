"""ctypes binding of libmapeval_hip.so (include/mapeval_hip.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAPEVAL_HIP_LIB: another build of the same library (e.g. the -DME_MME_STATS one, profiles/README.md); still no fallback
LIB_PATH = os.environ.get("MAPEVAL_HIP_LIB") or os.path.join(_HERE, "libmapeval_hip.so")

ME_OK = 0
ME_ERR_CAPACITY = -4
ME_SLOT_EST = 0
ME_SLOT_GT = 1
ME_GATE_LE_UNSQUARED = 0
ME_GATE_LT_SQUARED = 1
ME_FLAG_BORROW_DEVICE_INPUT = 1
ME_FLAG_MORTON_ORDER = 2
ME_SUITE_OVERLAP = 1
ME_SUITE_DEVICE_INPUT = 2
ME_SUITE_PIN_HOST_INPUT = 4
ME_LATTICE_BINS = 4096  # bins per axis of me_lattice_histograms_device

# every symbol include/mapeval_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "me_create", "me_destroy", "me_twin", "me_last_error", "me_version", "me_set_shard", "me_set_slab",
    "me_nn_unresolved", "me_nn_points", "me_nn_points_bounded", "me_nn_points_covered", "me_nn_cross_message", "me_nn_cross_answer", "me_nn_cross_patch", "me_nn_patch", "me_nn_fetch", "me_slab_points", "me_set_mme_result", "me_set_nn_result", "me_voxel_partials",
    "me_transform_points_device", "me_upload_slab_device", "me_halo_pack_device", "me_halo_pack_tagged_device", "me_lattice_histograms_device", "me_lattice_messages_device", "me_lattice_plan_device", "me_voxel_partial_rows_device", "me_voxel_merge_device",
    "me_upload_cloud", "me_upload_cloud_device", "me_cloud_size", "me_download_cloud", "me_voxel_downsample",
    "me_transform_cloud",
    "me_set_normals", "me_get_normals", "me_estimate_normals", "me_gicp_covariances", "me_get_covariances", "me_icp_lsq_sums",
    "me_nn1", "me_icp_p2p_sums", "me_render_distance", "me_render_entropy", "me_nn_stats", "me_nn_partial_sums", "me_nn_sigma_sums", "me_nn_finalize", "me_chamfer",
    "me_mme", "me_voxel_gaussians", "me_awd_scs", "me_w2_batch", "me_scs_table", "me_run_suite", "me_run_suite_from", "me_mme_fetch",
    "me_set_voxel_hint", "me_timers_enable", "me_timers_reset", "me_timer_get",
]


class NNPartial(C.Structure):
    _fields_ = [
        ("n_query", C.c_int64),
        ("n_corr", C.c_int64),
        ("n_inl", C.c_int64 * 5),
        ("sum_d", C.c_double * 5),
        ("sum_d2", C.c_double * 5),
        ("sum_sqrt_all", C.c_double),
    ]


class IcpSums(C.Structure):
    _fields_ = [
        ("n_corr", C.c_int64),
        ("n_source", C.c_int64),
        ("origin", C.c_double * 3),
        ("sum_p", C.c_double * 3),
        ("sum_q", C.c_double * 3),
        ("sum_pq", C.c_double * 9),
        ("sum_d2", C.c_double),
    ]


class IcpLsq(C.Structure):
    _fields_ = [
        ("n_corr", C.c_int64),
        ("n_source", C.c_int64),
        ("JTJ", C.c_double * 36),
        ("JTr", C.c_double * 6),
        ("r2", C.c_double),
        ("sum_d2", C.c_double),
    ]


class NNStatsOut(C.Structure):
    _fields_ = [
        ("n_src", C.c_int64),
        ("n_corr", C.c_int64),
        ("mean", C.c_double * 5),
        ("rmse", C.c_double * 5),
        ("fitness", C.c_double * 5),
        ("sigma", C.c_double * 5),
        ("number", C.c_double * 5),
        ("mean_nn_dist", C.c_double),
    ]


class SuiteParams(C.Structure):
    _fields_ = [
        ("icp_max_distance", C.c_double),
        ("gate_mode", C.c_int),
        ("trunc", C.c_double * 5),
        ("nn_radius", C.c_double),
        ("vmd_voxel_size", C.c_double),
        ("evaluate_mme", C.c_int),
        ("evaluate_gt_mme", C.c_int),
        ("min_pts", C.c_int),
        ("scs_radius", C.c_int),
    ]


class SuiteOut(C.Structure):
    _fields_ = [
        ("est_gt", NNStatsOut),
        ("gt_est", NNStatsOut),
        ("full_chamfer", C.c_double),
        ("mme_est", C.c_double),
        ("mme_gt", C.c_double),
        ("mme_est_valid", C.c_int64),
        ("mme_gt_valid", C.c_int64),
        ("awd", C.c_double),
        ("scs", C.c_double),
        ("n_w_voxels", C.c_int64),
        ("stage_ms", C.c_double * 8),
    ]


_lib = None


def load():
    """Load libmapeval_hip.so and declare prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C cloud_map_evaluation_amd/csrc).  There is no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses (host or device), passed as integers
    L.me_create.restype = C.c_void_p
    L.me_create.argtypes = [C.c_int, C.c_int]
    L.me_destroy.argtypes = [vp]
    L.me_destroy.restype = None
    L.me_twin.argtypes = [vp]
    L.me_twin.restype = C.c_void_p
    L.me_last_error.restype = C.c_char_p
    L.me_last_error.argtypes = [vp]
    L.me_version.restype = C.c_int
    L.me_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.me_set_slab.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double]
    L.me_nn_unresolved.argtypes = [vp, C.c_int, dp, dp, C.c_int64, C.POINTER(C.c_int64)]
    L.me_nn_points.argtypes = [vp, C.c_int, dp, C.c_int64, dp]
    L.me_nn_points_bounded.argtypes = [vp, C.c_int, dp, C.c_int64, dp]
    L.me_nn_points_covered.argtypes = [vp, C.c_int, dp, C.c_int64, dp, C.c_int, dp]
    L.me_nn_patch.argtypes = [vp, C.c_int, dp, C.c_int64]
    L.me_nn_cross_message.argtypes = [vp, dp, C.c_int64, C.c_int64, C.c_int64, dp]
    L.me_nn_cross_answer.argtypes = [vp, dp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, dp, C.c_double, dp]
    L.me_nn_cross_patch.argtypes = [vp, dp, C.c_int64, C.c_int]
    for f in ("me_nn_cross_message", "me_nn_cross_answer", "me_nn_cross_patch"):
        getattr(L, f).restype = C.c_int
    L.me_nn_fetch.argtypes = [vp, C.c_int, ip, dp]
    L.me_slab_points.argtypes = [vp, C.c_int, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.me_set_mme_result.argtypes = [vp, C.c_int, dp, vp]
    L.me_set_nn_result.argtypes = [vp, C.c_int, C.c_int, dp]
    L.me_voxel_partials.argtypes = [vp, C.c_int, C.c_double, ip, ip, dp, dp, C.POINTER(C.c_int64)]
    L.me_transform_points_device.argtypes = [vp, dp, C.c_int64, dp]
    L.me_upload_slab_device.argtypes = [vp, C.c_int, dp, C.c_int64, C.c_double]
    L.me_upload_slab_device.restype = C.c_int
    L.me_halo_pack_device.argtypes = [vp, dp, C.c_int64, C.c_int, dp, C.c_int, C.c_double, dp, C.c_int64, dp]
    L.me_halo_pack_tagged_device.argtypes = [vp, dp, C.c_int64, C.c_int, dp, C.c_int, C.c_double, dp, dp, C.c_int64, C.c_int64, dp]
    L.me_halo_pack_tagged_device.restype = C.c_int
    L.me_lattice_histograms_device.argtypes = [vp, dp, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]
    L.me_lattice_histograms_device.restype = C.c_int
    L.me_lattice_messages_device.argtypes = [vp, dp, C.c_int64, dp, C.c_int64, C.c_int, C.c_int, vp]
    L.me_lattice_messages_device.restype = C.c_int
    L.me_lattice_plan_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int64)]
    L.me_lattice_plan_device.restype = C.c_int
    L.me_voxel_partial_rows_device.argtypes = [vp, C.c_int, C.c_double, dp, C.c_int64, C.POINTER(C.c_int64)]
    L.me_voxel_merge_device.argtypes = [vp, C.c_int, C.c_double, dp, C.c_int64]
    for f in ("me_transform_points_device", "me_upload_slab_device", "me_halo_pack_device", "me_voxel_partial_rows_device", "me_voxel_merge_device"):
        getattr(L, f).restype = C.c_int
    L.me_set_voxel_hint.argtypes = [vp, C.c_double]
    L.me_set_voxel_hint.restype = C.c_int
    L.me_voxel_downsample.argtypes = [vp, C.c_int, C.c_double, C.POINTER(C.c_int64)]
    L.me_transform_cloud.argtypes = [vp, C.c_int, dp]
    L.me_upload_cloud.argtypes = [vp, C.c_int, dp, C.c_int64, dp, C.c_double]
    L.me_upload_cloud_device.argtypes = [vp, C.c_int, dp, C.c_int64, dp, C.c_double]
    L.me_cloud_size.restype = C.c_int64
    L.me_cloud_size.argtypes = [vp, C.c_int]
    L.me_download_cloud.argtypes = [vp, C.c_int, dp]
    L.me_nn1.argtypes = [vp, C.c_int, C.c_int, ip, dp]
    L.me_icp_p2p_sums.argtypes = [vp, C.c_int, C.c_double, C.POINTER(IcpSums)]
    L.me_icp_p2p_sums.restype = C.c_int
    L.me_set_normals.argtypes = [vp, C.c_int, dp]
    L.me_get_normals.argtypes = [vp, C.c_int, dp]
    L.me_estimate_normals.argtypes = [vp, C.c_int, C.c_int, dp, ip, dp]
    L.me_gicp_covariances.argtypes = [vp, C.c_int, C.c_double, dp]
    L.me_get_covariances.argtypes = [vp, C.c_int, dp]
    L.me_get_covariances.restype = C.c_int
    L.me_icp_lsq_sums.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.POINTER(IcpLsq)]
    for f in ("me_set_normals", "me_get_normals", "me_estimate_normals", "me_gicp_covariances", "me_icp_lsq_sums"):
        getattr(L, f).restype = C.c_int
    L.me_render_distance.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int, vp, vp]
    L.me_render_distance.restype = C.c_int
    L.me_render_entropy.argtypes = [vp, C.c_int, vp, vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]
    L.me_render_entropy.restype = C.c_int
    L.me_nn_stats.argtypes = [vp, C.c_int, C.c_double, C.c_int, dp, C.POINTER(NNStatsOut)]
    L.me_nn_partial_sums.argtypes = [vp, C.c_int, C.c_double, C.c_int, dp, C.POINTER(NNPartial)]
    L.me_nn_sigma_sums.argtypes = [vp, C.c_int, C.c_double, C.c_int, dp, dp]
    L.me_nn_finalize.restype = None
    L.me_nn_finalize.argtypes = [C.POINTER(NNPartial), dp, C.c_int64, C.POINTER(NNStatsOut)]
    L.me_chamfer.argtypes = [vp, C.POINTER(C.c_double)]
    L.me_mme.argtypes = [vp, C.c_int, C.c_double, C.c_int, dp, dp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.me_voxel_gaussians.argtypes = [vp, C.c_int, C.c_double, ip, ip, dp, dp, dp, C.POINTER(C.c_int64)]
    L.me_awd_scs.argtypes = [vp, C.c_double, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                             C.POINTER(C.c_double), dp]
    L.me_run_suite.argtypes = [vp, C.POINTER(SuiteParams), C.POINTER(SuiteOut)]
    L.me_run_suite_from.argtypes = [vp, dp, C.c_int64, dp, C.c_int64, dp, C.POINTER(SuiteParams), C.c_int, C.POINTER(SuiteOut)]
    L.me_mme_fetch.argtypes = [vp, C.c_int, dp, dp]
    L.me_w2_batch.argtypes = [vp, dp, dp, ip, dp, dp, ip, C.c_int64, dp]
    L.me_scs_table.argtypes = [vp, ip, dp, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    L.me_timers_enable.argtypes = [vp, C.c_int]
    L.me_timers_reset.argtypes = [vp]
    L.me_timer_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    for f in ("me_voxel_downsample", "me_transform_cloud", "me_set_slab", "me_nn_unresolved", "me_nn_points", "me_nn_points_bounded", "me_nn_points_covered", "me_nn_patch", "me_nn_fetch", "me_slab_points", "me_set_mme_result", "me_set_nn_result", "me_voxel_partials", "me_set_shard", "me_upload_cloud", "me_upload_cloud_device", "me_download_cloud", "me_nn1", "me_icp_p2p_sums", "me_render_distance", "me_render_entropy", "me_nn_stats",
              "me_nn_partial_sums", "me_nn_sigma_sums", "me_chamfer", "me_mme", "me_voxel_gaussians", "me_awd_scs",
              "me_run_suite", "me_run_suite_from", "me_mme_fetch", "me_w2_batch", "me_scs_table", "me_timers_enable", "me_timers_reset", "me_timer_get"):
        getattr(L, f).restype = C.c_int
    _lib = L
    return L
