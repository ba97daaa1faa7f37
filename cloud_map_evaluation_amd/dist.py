"""Multi-GPU driver of the metric suite: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Sharding ("primary" variant): every rank holds both clouds and builds both indices (the
reference cloud must be complete on every rank: CD's nearest neighbour is unbounded, map_eval.cpp:1398-1431);
the per-point passes (1-NN both directions, MME) process only the rank's slab of the Morton-sorted query order
(me_set_shard) and return raw partial sums (me_nn_partial / sum_H, n_valid).  The data path needs exactly two
collectives per suite: one all-reduce (sum) of a 38-double vector of partials, and one all-reduce of the 2 x 5
sigma numerators (the second pass of map_eval.cpp:1132-1138 needs the global means first).  The voxel / AWD / SCS
stage is O(V) and replicated (no exchange).

`suite_step` is engine-agnostic (duck-typed): tests drive it on CPU (gloo, world_size 2) with a stand-in engine.
"""
from __future__ import annotations

import numpy as np

ME_SLOT_EST, ME_SLOT_GT = 0, 1
ME_GATE_LE_UNSQUARED = 0

# layout of the all-reduced vector: per direction [n_corr, n_inl[5], sum_d[5], sum_d2[5], sum_sqrt_all] = 17 doubles
_DIR = 17
VEC_LEN = 2 * _DIR + 4  # + [mme_est_sum, mme_est_valid, mme_gt_sum, mme_gt_valid]


def shard_range(n: int, rank: int, world: int):
    """The slab of a length-n pass owned by `rank` (same formula as me_ctx::shard_range)."""
    return n * rank // world, n * (rank + 1) // world


def pack_partials(parts, mme_est, mme_gt) -> np.ndarray:
    """parts: two me_nn_partial-like objects (est->gt, gt->est); mme_*: (sum_H, n_valid).  Counts < 2^53 are exact."""
    vec = []
    for pp in parts:
        vec += [pp.n_corr] + list(pp.n_inl) + list(pp.sum_d) + list(pp.sum_d2) + [pp.sum_sqrt_all]
    vec += [mme_est[0], mme_est[1], mme_gt[0], mme_gt[1]]
    return np.asarray(vec, dtype=np.float64)


def all_reduce_sum(vec: np.ndarray, dist, device) -> np.ndarray:
    """Sum over ranks (no-op without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec
    import torch

    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()


def direction_stats(vec: np.ndarray, i: int, sigma_num: np.ndarray, n_src: int) -> dict:
    """getDiffRegResultWithCorrespondence's final arithmetic (map_eval.cpp:1125-1144) on all-reduced sums."""
    o = i * _DIR
    C = vec[o]
    nan5 = np.full(5, np.nan)
    return dict(
        n_corr=int(C),
        number=vec[o + 1:o + 6].copy(),
        mean=vec[o + 6:o + 11] / C if C > 0 else nan5,
        rmse=np.sqrt(vec[o + 11:o + 16] / C) if C > 0 else nan5,
        fitness=vec[o + 1:o + 6] / n_src,
        sigma=np.sqrt(sigma_num / C) if C > 0 else nan5,
        mean_nn=vec[o + 16] / n_src,
    )


def suite_step(eng, dist, device, est, gt, P, evaluate_gt_mme: bool = True, upload: bool = True):
    """One full pass of the hot path (what MapEval::process runs between load and save, map_eval.cpp:52-85).

    eng: Engine (or a stand-in with the same methods) already sharded with set_shard(rank, world).
    est / gt: clouds (host arrays or device tensors); P: Param.  Returns the dict of scalars, identical on all ranks.
    """
    if upload:
        eng.upload(ME_SLOT_EST, est, T=np.asarray(P.initial_matrix_, dtype=np.float64), cell_size=P.nn_radius_)
        eng.upload(ME_SLOT_GT, gt, cell_size=P.nn_radius_)
    n_e, n_g = eng.size(ME_SLOT_EST), eng.size(ME_SLOT_GT)
    # --- MME (map_eval.cpp:56): k >= 10 for the estimated map (:1675), k >= 5 for the ground truth (:1458) ---
    if P.evaluate_mme_:
        m = eng.mme(ME_SLOT_EST, P.nn_radius_, 10, per_point=False)
        m_e = (m[4], m[3])
        if evaluate_gt_mme:
            m = eng.mme(ME_SLOT_GT, P.nn_radius_, 5, per_point=False)
            m_g = (m[4], m[3])
        else:
            m_g = (0.0, 0)
    else:
        m_e = m_g = (0.0, 0)
    # --- AC / COM / CD (map_eval.cpp:76, :1194): both directions, partial sums over the rank's slab ---
    parts = []
    for q, r in ((ME_SLOT_EST, ME_SLOT_GT), (ME_SLOT_GT, ME_SLOT_EST)):
        eng.nn1(q, r, fetch=False)
        parts.append(eng.nn_partial_sums(q, P.icp_max_distance_, ME_GATE_LE_UNSQUARED, P.trunc_dist_))
    vec = all_reduce_sum(pack_partials(parts, m_e, m_g), dist, device)  # collective #1
    # second pass: sigma needs the global mean of every threshold (map_eval.cpp:1132-1138)
    sig_local = []
    for i, q in enumerate((ME_SLOT_EST, ME_SLOT_GT)):
        C = vec[i * _DIR]
        mean = vec[i * _DIR + 6:i * _DIR + 11] / C if C > 0 else np.zeros(5)
        sig_local.append(eng.nn_sigma_sums(q, P.icp_max_distance_, ME_GATE_LE_UNSQUARED, mean))
    sig = all_reduce_sum(np.concatenate(sig_local), dist, device)  # collective #2
    s_eg = direction_stats(vec, 0, sig[:5], n_e)
    s_ge = direction_stats(vec, 1, sig[5:], n_g)
    o = 2 * _DIR
    mme_est = vec[o] / vec[o + 1] if vec[o + 1] > 0 else 0.0      # (:1720-1724)
    mme_gt = vec[o + 2] / vec[o + 3] if vec[o + 3] > 0 else 0.0
    # --- AWD / SCS (map_eval.cpp:85): O(V) voxel tables, replicated on every rank ---
    v = eng.calculateVMD(P.vmd_voxel_size_, rows=False)
    return dict(est_gt=s_eg, gt_est=s_ge, ac=s_eg["rmse"], com=s_eg["fitness"], cd=s_eg["mean_nn"] + s_ge["mean_nn"],
                mme_est=mme_est, mme_gt=mme_gt, mme_valid=int(vec[o + 1]), awd=v["awd"], scs=v["scs"], n_w=v["n_rows"],
                n_est=n_e, n_gt=n_g)
