"""Spatial slab mode (multi-GPU design) exercised on ONE GPU.

1. slab primitives, ranks emulated one after another: MME partials, voxel partials (Chan merge) and the local 1-NN +
   cross-rank resolve add up to / reproduce the single-context result exactly (counts) or to 1e-12 (sums);
2. the real driver (cloud_map_evaluation_amd.dist.suite_step_slab) with TWO processes sharing this GPU and gloo
   collectives on the CPU — the same code path the 8-GPU run takes with RCCL — against the CPU oracle.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)


def _scene(n=120_000):
    from cloud_map_evaluation_amd import synth

    est, gt = synth.campus_pair(n, density=2500.0, seed=5, origin=(100.0, -50.0, 3.0))
    return est.numpy(), gt.numpy()


def test_slab_primitives_sum_to_whole():
    import torch

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Engine

    est, gt = _scene()
    # a few far-away est points: their nearest GT point lies in another slab -> cross-rank step
    est = np.concatenate([est, est[:200] + np.array([3.0, 0.0, 25.0])])
    world = 4
    with Engine(0) as eng:
        eng.upload(0, est, cell_size=0.1)
        eng.upload(1, gt, cell_size=0.1)
        whole_mme = eng.mme(0, 0.1, 10)
        _, d2_whole = eng.nn1(0, 1)
        whole_tab = eng.voxel_gaussians(0, 1.0)
        # --- emulate the ranks ---
        mme_s, mme_c, sum_sqrt, n_q = 0.0, 0, 0.0, 0
        rows_all = []
        unresolved_total = 0
        for rank in range(world):
            axis, lo, hi = medist.slab_bounds(torch.from_numpy(gt), rank, world)
            eng.set_slab(axis, lo, hi, 0.5)
            eng.upload(0, est, cell_size=0.1)
            eng.upload(1, gt, cell_size=0.1)
            m = eng.mme(0, 0.1, 10, per_point=False)
            mme_s += m[4]
            mme_c += m[3]
            eng.nn1(0, 1, fetch=False)
            open_q = eng.nn_unresolved(0)
            unresolved_total += len(open_q)
            if len(open_q):
                # "other ranks" answer: emulate with a slab-free context holding the whole GT cloud
                with Engine(0) as full:
                    full.upload(1, gt, cell_size=0.1)
                    eng.nn_patch(0, full.nn_points(1, open_q))
            p = eng.nn_partial_sums(0, -1.0, 0, TRUNC)
            sum_sqrt += p.sum_sqrt_all
            n_q += p.n_corr  # gate < 0: every owned query is a correspondence
            k, n, mu, m2 = eng.voxel_partials(0, 1.0)
            rows_all.append(np.concatenate([k.astype(float), n[:, None].astype(float), mu, m2.reshape(-1, 9)], 1))
        eng.set_slab(-1)
    assert n_q == len(est)                                    # every query owned by exactly one rank
    assert unresolved_total >= 100                            # the lifted points needed the cross-rank step
    assert mme_c == whole_mme[3]                              # bit-exact valid count across slabs
    np.testing.assert_allclose(mme_s, whole_mme[4], rtol=1e-12)
    np.testing.assert_allclose(sum_sqrt, np.sqrt(d2_whole).sum(), rtol=1e-12)
    keys, n, mu, sig = medist.merge_voxel_partials(np.concatenate(rows_all))
    wk, wn, wmu, wsig, _ = whole_tab
    assert np.array_equal(keys, wk) and np.array_equal(n, wn)  # same voxels, same populations
    np.testing.assert_allclose(mu, wmu, rtol=1e-13)
    from tests._tol import assert_sigma_close

    assert_sigma_close(sig, wsig)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, est, gt, T, q, overlap, backend="gloo"):
    import torch
    import torch.distributed as dist

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Engine, Param

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        os.environ["ME_FORCE_COLLECTIVES"] = "1"  # one rank, but every collective of the step goes through RCCL
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=1.0, initial_matrix_=T)
        comm = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
        with Engine(0) as eng:
            res = medist.suite_step_slab(eng, dist, comm, est, gt, P, rank, world, halo=0.5, overlap=overlap)
        q.put((rank, {k: (v if not isinstance(v, dict) else {kk: np.asarray(vv) for kk, vv in v.items()}) for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", [False, True])  # True: the second lane (me_twin) indexes GT + builds the voxel partials
def test_two_process_slab_suite_matches_oracle(overlap):
    import torch.multiprocessing as mp

    import oracle

    est, gt = _scene(100_000)
    est = np.concatenate([est, est[:150] + np.array([2.0, 0.0, 30.0])])
    T = np.eye(4)
    T[:3, 3] = [0.003, -0.002, 0.001]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, est, gt, T, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    est_t = oracle.transform(est, T)
    o_eg = oracle.reg_stats(est_t, gt, 1.0, 0, TRUNC)
    o_ge = oracle.reg_stats(gt, est_t, 1.0, 0, TRUNC)
    o_me = oracle.mme(est_t, 0.1, 10)
    o_mg = oracle.mme(gt, 0.1, 5)
    o_v = oracle.awd_scs(oracle.VoxelMap(gt, 1.0), oracle.VoxelMap(est_t, 1.0))
    for rank in (0, 1):
        r = results[rank]
        assert r["n_cross_rank_queries"] >= 100
        for got, exp in ((r["est_gt"], o_eg), (r["gt_est"], o_ge)):
            assert got["n_corr"] == exp.n_corr
            assert np.array_equal(got["number"], exp.number)          # bit-exact inlier counts
            for k in ("mean", "rmse", "sigma"):
                np.testing.assert_allclose(got[k], getattr(exp, k), rtol=1e-9)
        np.testing.assert_allclose(r["cd"], oracle.chamfer(est_t, gt), rtol=1e-9)
        assert r["mme_valid"] == o_me[3]
        np.testing.assert_allclose(r["mme_est"], o_me[0], rtol=1e-9)
        np.testing.assert_allclose(r["mme_gt"], o_mg[0], rtol=1e-9)
        assert r["n_w"] == len(o_v["rows"])
        np.testing.assert_allclose(r["awd"], o_v["awd"], rtol=1e-9)
        np.testing.assert_allclose(r["scs"], o_v["scs"], rtol=1e-9)


@pytest.mark.timeout(600)
def test_one_rank_step_through_rccl_matches_the_plain_step():
    """There is one GPU here, so the RCCL calls of the slab step cannot be exercised across ranks; a one-rank `nccl`
    process group with ME_FORCE_COLLECTIVES=1 still sends every all-reduce / all-gather of the step through RCCL on
    device tensors (int64 MAX, float64 SUM, padded float64 gathers): the result must equal the same step without a
    process group."""
    import torch
    import torch.multiprocessing as mp

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Engine, Param

    est, gt = _scene(60_000)
    T = np.eye(4)
    T[:3, 3] = [0.003, -0.002, 0.001]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), est, gt, T, q, True, "nccl"))
    p.start()
    rank, r = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0 and rank == 0
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=1.0, initial_matrix_=T)
    with Engine(0) as eng:
        ref = medist.suite_step_slab(eng, None, torch.device("cpu"), est, gt, P, 0, 1, halo=0.5, overlap=True)
    for k in ("cd", "mme_est", "mme_gt", "awd", "scs"):
        assert r[k] == ref[k], k
    for d in ("est_gt", "gt_est"):
        for k in ("mean", "rmse", "sigma", "number", "fitness"):
            assert np.array_equal(np.asarray(r[d][k]), np.asarray(ref[d][k])), (d, k)
    assert r["n_w"] == ref["n_w"] and r["mme_valid"] == ref["mme_valid"]


@pytest.mark.parametrize("prefiltered", [False, True])
def test_per_point_outputs_in_slab_mode_cover_the_whole_cloud(prefiltered):
    """Round 3: me_mme / me_nn1 return per-point arrays under a slab — one entry per point the context
    holds, me_slab_points says which uploaded point each entry is and whether the rank owns it.  The owned entries of the
    ranks, put together, are the single-context arrays: valid flags and squared distances bit for bit, entropies to 1e-12 (the
    candidate streams are summed in a different order); what map_entropy.pcd / raw_rendered_dis_map.pcd need
    (map_eval.cpp:485-495, 686-736)."""
    import torch

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Engine

    est, gt = _scene(90_000)
    world, halo = 3, 1.0
    with Engine(0) as eng:
        eng.upload(0, est, cell_size=0.1)
        eng.upload(1, gt, cell_size=0.1)
        _, ent_w, val_w, _, _ = eng.mme(0, 0.1, 10)
        idx_w, d2_w = eng.nn1(0, 1)
        ent = np.full(len(est), np.nan)
        val = np.zeros(len(est), np.uint8)
        d2 = np.full(len(est), np.nan)
        nn_xyz = np.full((len(est), 3), np.nan)
        seen = np.zeros(len(est), np.int32)
        for rank in range(world):
            axis, lo, hi = medist.slab_bounds(torch.from_numpy(gt), rank, world)
            eng.set_slab(axis, lo, hi, halo)
            if prefiltered:  # what the halo exchange delivers: exactly the slab + halo, uploaded without the filter pass
                ke = np.nonzero((est[:, axis] >= lo - halo) & (est[:, axis] < hi + halo))[0]
                kg = np.nonzero((gt[:, axis] >= lo - halo) & (gt[:, axis] < hi + halo))[0]
                eng.upload_slab(0, torch.from_numpy(est[ke]).cuda(), cell_size=0.1)
                eng.upload_slab(1, torch.from_numpy(gt[kg]).cuda(), cell_size=0.1)
            else:
                ke, kg = np.arange(len(est)), np.arange(len(gt))
                eng.upload(0, est, cell_size=0.1)
                eng.upload(1, gt, cell_size=0.1)
            orig, owned = eng.slab_points(0)
            orig_g, _ = eng.slab_points(1)
            assert len(orig) == eng.size(0) and np.all(np.diff(orig) > 0)
            coord = est[ke[orig], axis]
            assert np.array_equal(owned, (coord >= lo) & (coord < hi))
            _, e_r, v_r, nv, _ = eng.mme(0, 0.1, 10)
            assert nv == int(v_r.sum()) and not v_r[~owned].any() and not e_r[~owned].any()
            i_r, d_r = eng.nn1(0, 1)
            assert np.all(d_r[~owned] == -1.0) and np.all(i_r[~owned] == -1)
            open_q = eng.nn_unresolved(0)
            if len(open_q):
                with Engine(0) as full:
                    full.upload(1, gt, cell_size=0.1)
                    eng.nn_patch(0, full.nn_points(1, open_q))
                i_r, d_r = eng.nn_fetch(0)
            g = ke[orig[owned]]
            seen[g] += 1
            ent[g], val[g], d2[g] = e_r[owned], v_r[owned], d_r[owned]
            nn_xyz[g] = gt[kg[orig_g[np.maximum(i_r[owned], 0)]]]
        eng.set_slab(-1)
    assert np.all(seen == 1), "every point is owned by exactly one rank"
    assert np.array_equal(val, val_w) and np.array_equal(d2, d2_w)
    np.testing.assert_allclose(ent, ent_w, rtol=1e-12, atol=0)
    # the local neighbour index is the global one wherever the local search was final
    dd = est - nn_xyz
    local = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
    assert (local == d2_w).mean() > 0.99 and np.all(local >= d2_w)


def test_slab_upload_rejects_points_outside_the_declared_region():
    """me_upload_slab_device trusts that the exchange delivered exactly slab + halo; a caller whose cuts or halo differ from
    me_set_slab's gets a loud error, not silently incomplete neighbourhoods near the faces."""
    import torch

    from cloud_map_evaluation_amd.engine import Engine, MapEvalError

    est, _ = _scene(20_000)
    axis = 0
    lo, hi = np.quantile(est[:, axis], [0.3, 0.6])
    with Engine(0) as eng:
        eng.set_slab(axis, lo, hi, 0.5)
        inside = est[(est[:, axis] >= lo - 0.5) & (est[:, axis] < hi + 0.5)]
        eng.upload_slab(0, torch.from_numpy(inside).cuda(), cell_size=0.1)  # exactly the region: accepted
        assert eng.size(0) == len(inside)
        wider = est[(est[:, axis] >= lo - 0.8) & (est[:, axis] < hi + 0.5)]  # an exchange run with a wider halo
        assert len(wider) > len(inside)
        with pytest.raises(MapEvalError, match="outside"):
            eng.upload_slab(0, torch.from_numpy(wider).cuda(), cell_size=0.1)
        eng.set_slab(-1)


def test_filtered_upload_then_prefiltered_upload_on_the_same_slot_is_the_identity_again():
    """the filtered slab upload clears Cloud::slab_identity; a later me_upload_slab_device on the same
    context and slot must set it back, or me_slab_points answers with the previous upload's (stale, possibly shorter)
    orig-index table."""
    import torch

    from cloud_map_evaluation_amd.engine import Engine

    est, _ = _scene(40_000)
    axis = 0
    lo, hi = np.quantile(est[:, axis], [0.2, 0.5])
    halo = 0.5
    with Engine(0) as eng:
        eng.set_slab(axis, lo, hi, halo)
        eng.upload(0, est, cell_size=0.1)  # filtered: keeps slab + halo of the whole cloud, orig != identity
        orig_f, _ = eng.slab_points(0)
        assert len(orig_f) < len(est) and not np.array_equal(orig_f, np.arange(len(orig_f)))
        # a LARGER prefiltered piece on the same slot (the stale table would be read out of bounds)
        lo2, hi2 = np.quantile(est[:, axis], [0.1, 0.9])
        eng.set_slab(axis, lo2, hi2, halo)
        inside = est[(est[:, axis] >= lo2 - halo) & (est[:, axis] < hi2 + halo)]
        assert len(inside) > len(orig_f)
        eng.upload_slab(0, torch.from_numpy(inside).cuda(), cell_size=0.1)
        orig_p, owned_p = eng.slab_points(0)
        assert np.array_equal(orig_p, np.arange(len(inside)))
        assert np.array_equal(owned_p, (inside[:, axis] >= lo2) & (inside[:, axis] < hi2))
        eng.set_slab(-1)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_nn_points_covered_gives_the_bounded_answer(axis):
    """me_nn_points_covered (round 4): queries whose owner has searched a band of one axis completely arrive with the nearest
    squared distance found in the band; the answering context only looks outside the band and must return exactly
    min(bound, nearest squared distance to ALL its points) — what me_nn_points_bounded returns, bit for bit."""
    import oracle
    import torch

    from cloud_map_evaluation_amd.engine import Engine

    rng = np.random.default_rng(3 + axis)
    _, gt = _scene(150_000)
    if axis == 2:  # (the scene is flat: tilt it so that a band of z holds a part of it)
        gt = gt @ np.array([[1.0, 0, 0], [0, 0.8, -0.6], [0, 0.6, 0.8]])
    lo, hi = np.quantile(gt[:, axis], [0.35, 0.6])
    inside = (gt[:, axis] >= lo) & (gt[:, axis] < hi)
    assert 1000 < inside.sum() < len(gt) - 1000
    # queries inside the band: surface points with noise, far outliers (metres off: balls that reach out of the band), points
    # next to the band's ends (their neighbour is often just outside)
    base = gt[inside][rng.choice(inside.sum(), 3000, replace=False)]
    q = np.concatenate([base + rng.normal(0, 0.03, base.shape),
                        base[:800] + rng.normal(0, 1.0, (800, 3)) * np.array([1.0, 1.0, 8.0]),
                        base[:400] + rng.normal(0, 6.0, (400, 3))])
    q[:, axis] = np.clip(q[:, axis], lo, np.nextafter(hi, -np.inf))  # owned by the band's rank
    q[-200:, axis] = np.where(rng.random(200) < 0.5, lo + rng.random(200) * 0.05, hi - 0.05 + rng.random(200) * 0.0499)
    bound = oracle.nn1(np.ascontiguousarray(gt[inside]), q)[1]     # what the owner found in its band
    expect = oracle.nn1(np.ascontiguousarray(gt), q)[1]            # the exact answer over everything
    assert (expect < bound).sum() > 50 and (expect == bound).sum() > 1000  # both kinds of query are present
    dev = torch.device("cuda", 0)
    with Engine(0) as eng:
        eng.upload(1, gt, cell_size=0.1)
        qd, bd = torch.from_numpy(q).to(dev), torch.from_numpy(bound).to(dev)
        cov = torch.tensor([[lo, hi]], dtype=torch.float64).expand(len(q), 2).contiguous()
        got = eng.nn_points(1, qd, bound=bd, covered=cov, axis=axis).cpu().numpy()
        plain = eng.nn_points(1, qd, bound=bd).cpu().numpy()
        # slots that need no answer (bound -1) come back unchanged; an empty band [inf, -inf) is the plain bounded search
        neg = eng.nn_points(1, qd[:64], bound=torch.full((64,), -1.0, dtype=torch.float64, device=dev), covered=cov[:64], axis=axis).cpu().numpy()
        none = torch.tensor([[np.inf, -np.inf]], dtype=torch.float64).expand(len(q), 2).contiguous()
        got_none = eng.nn_points(1, qd, bound=bd, covered=none, axis=axis).cpu().numpy()
    np.testing.assert_array_equal(plain, expect)
    np.testing.assert_array_equal(got, expect)
    np.testing.assert_array_equal(got_none, expect)
    assert (neg == -1.0).all()


def test_cross_rank_calls_reproduce_the_whole_cloud_search():
    """me_nn_cross_message / _answer / _patch (round 5): two contexts in one process play the two ranks of a slab job; the messages are
    'all-gathered' by stacking them, the answer blocks 'min-reduced' with torch.minimum.  After the patch, every owned query's squared
    distance must be the WHOLE-cloud oracle's, bit for bit, in both directions — and equal to what the piecewise path
    (me_nn_unresolved / me_nn_points_covered / me_nn_patch) gives."""
    import oracle
    import torch

    from cloud_map_evaluation_amd.engine import Engine

    est, gt = _scene(160_000)
    rng = np.random.default_rng(9)
    est = np.concatenate([est, est[rng.choice(len(est), 1500, replace=False)] + rng.normal(0, 2.5, (1500, 3))])  # far outliers: open queries
    axis, halo, cap = 0, 1.0, 4096
    cut = float(np.median(gt[:, axis]))
    cuts = [-np.inf, cut, np.inf]
    want = {0: oracle.nn1(gt, est)[1], 1: oracle.nn1(est, gt)[1]}  # query slot -> exact squared distances over the whole reference
    engs = [Engine(0), Engine(0)]
    try:
        msgs, counts = [], []
        for r, e in enumerate(engs):
            e.set_slab(axis, cuts[r], cuts[r + 1], halo)
            e.upload(0, est, cell_size=0.1)
            e.upload(1, gt, cell_size=0.1)
            e.nn1(0, 1, fetch=False)
            e.nn1(1, 0, fetch=False)
            c = [e.nn_unresolved_count(0), e.nn_unresolved_count(1)]
            m, c2 = e.nn_cross_message(cap, e.size(0), e.size(1))
            assert c2 == c and m.shape == (1 + 2 * cap, 4)
            head = m[0].cpu().numpy()
            assert list(head[:2]) == c and (m[1 + c[0]:1 + cap, 3] == -1.0).all() and (m[1 + cap + c[1]:, 3] == -1.0).all()
            msgs.append(m)
            counts.append(c)
        assert sum(c[0] for c in counts) > 100, "the scene was meant to leave open queries"
        gathered = torch.stack(msgs)                                   # the all-gather
        blocks = [e.nn_cross_answer(gathered, cap, r, 3, axis, cuts, halo) for r, e in enumerate(engs)]
        # own slots keep the owner's bound, padding stays -1
        for r in range(2):
            assert torch.equal(blocks[r][r, 1:], gathered[r, 1:, 3])
        reduced = torch.minimum(blocks[0], blocks[1])                  # the all-reduce MIN
        for r, e in enumerate(engs):
            e.nn_cross_patch(reduced, cap, r)
            for q in (0, 1):
                _, d2 = e.nn_fetch(q)
                orig, owned = e.slab_points(q)
                sel = owned.astype(bool)
                np.testing.assert_array_equal(d2[sel], want[q][orig[sel]])
        # dir_mask 0: a direction nobody else has open queries in is copied through
        thru = engs[0].nn_cross_answer(gathered, cap, 0, 0, axis, cuts, halo)
        assert torch.equal(thru[:, 1:], gathered[:, 1:, 3])
    finally:
        for e in engs:
            e.close()
