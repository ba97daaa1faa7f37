"""N > 1 path on CPU: world_size-2 gloo run of cloud_map_evaluation_amd.dist.suite_step.

The GPU engine is replaced by a stand-in that produces, per rank, the raw partial sums of ITS SLAB (computed with
the CPU oracle — test infrastructure); everything under test is product code: slab partition, packing, the two
all-reduces, and the final arithmetic (map_eval.cpp:1125-1144).  The reduced result must equal the oracle's
single-process result: bit-exact counts, fp within 1e-12.
"""
import os
import socket
import types

import numpy as np
import pytest

TRUNC = (0.2, 0.1, 0.08, 0.05, 0.01)


class OracleShardEngine:
    """Duck-typed stand-in for cloud_map_evaluation_amd.engine.Engine restricted to one slab."""

    def __init__(self):
        self.rank, self.world = 0, 1
        self.cloud = {}
        self.d2 = {}

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world

    def upload(self, slot, xyz, T=None, cell_size=0.0):
        import oracle

        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        self.cloud[slot] = oracle.transform(xyz, T) if T is not None else xyz

    def size(self, slot):
        return len(self.cloud[slot])

    def _slab(self, n):
        from cloud_map_evaluation_amd.dist import shard_range

        return slice(*shard_range(n, self.rank, self.world))

    def mme(self, slot, radius, min_k, per_point=False):
        import oracle

        _, ent, val, _, _ = oracle.mme(self.cloud[slot], radius, min_k)
        s = self._slab(len(ent))
        return 0.0, None, None, int(val[s].sum()), float(ent[s].sum())

    def nn1(self, q, r, fetch=False):
        import oracle

        self.d2[q] = oracle.nn1(self.cloud[r], self.cloud[q])[1]
        return None, None

    def _gate(self, d2, gate, mode):
        if gate < 0:
            return np.ones_like(d2, bool)
        return d2 <= gate if mode == 0 else d2 < gate * gate

    def nn_partial_sums(self, q, gate, mode, trunc):
        d2 = self.d2[q][self._slab(len(self.d2[q]))]
        keep = self._gate(d2, gate, mode)
        d = np.sqrt(d2)
        out = types.SimpleNamespace(n_query=len(d2), n_corr=int(keep.sum()), n_inl=[], sum_d=[], sum_d2=[],
                                    sum_sqrt_all=float(d.sum()))
        for t in trunc:
            inl = keep & (d <= t)
            out.n_inl.append(int(inl.sum()))
            out.sum_d.append(float(d[inl].sum()))
            out.sum_d2.append(float(d2[inl].sum()))
        return out

    def nn_sigma_sums(self, q, gate, mode, mean):
        d2 = self.d2[q][self._slab(len(self.d2[q]))]
        d = np.sqrt(d2[self._gate(d2, gate, mode)])
        return np.array([((d - m) ** 2).sum() for m in mean])

    def calculateVMD(self, vs, rows=False):
        import oracle

        r = oracle.awd_scs(oracle.VoxelMap(self.cloud[1], vs), oracle.VoxelMap(self.cloud[0], vs))
        return dict(awd=r["awd"], scs=r["scs"], n_rows=len(r["rows"]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, est, gt, T, q):
    import torch
    import torch.distributed as dist

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Param

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleShardEngine()
        eng.set_shard(rank, world)
        P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=0.5, initial_matrix_=T)
        res = medist.suite_step(eng, dist, torch.device("cpu"), est, gt, P, evaluate_gt_mme=True)
        q.put((rank, {k: (v if not isinstance(v, dict) else {kk: np.asarray(vv) for kk, vv in v.items()}) for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_suite_equals_single_process_oracle():
    import torch.multiprocessing as mp

    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(12000, seed=21)
    est, gt = est.numpy() * 0.5, gt.numpy()[:11000] * 0.5
    T = np.eye(4)
    T[:3, 3] = [0.004, -0.003, 0.002]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, est, gt, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    est_t = oracle.transform(est, T)
    o_eg = oracle.reg_stats(est_t, gt, 1.0, 0, TRUNC)
    o_ge = oracle.reg_stats(gt, est_t, 1.0, 0, TRUNC)
    o_me = oracle.mme(est_t, 0.1, 10)
    o_mg = oracle.mme(gt, 0.1, 5)
    o_v = oracle.awd_scs(oracle.VoxelMap(gt, 0.5), oracle.VoxelMap(est_t, 0.5))
    for rank in (0, 1):  # every rank ends with the same, complete answer
        r = results[rank]
        for got, exp in ((r["est_gt"], o_eg), (r["gt_est"], o_ge)):
            assert got["n_corr"] == exp.n_corr
            assert np.array_equal(got["number"], exp.number)  # bit-exact inlier counts after the all-reduce
            assert np.array_equal(got["fitness"], exp.fitness)
            for k in ("mean", "rmse", "sigma"):
                np.testing.assert_allclose(got[k], getattr(exp, k), rtol=1e-12)
        np.testing.assert_allclose(r["cd"], oracle.chamfer(est_t, gt), rtol=1e-12)
        assert r["mme_valid"] == o_me[3]
        np.testing.assert_allclose(r["mme_est"], o_me[0], rtol=1e-12)
        np.testing.assert_allclose(r["mme_gt"], o_mg[0], rtol=1e-12)
        np.testing.assert_allclose(r["awd"], o_v["awd"], rtol=1e-12)
        np.testing.assert_allclose(r["scs"], o_v["scs"], rtol=1e-12)


def test_shard_ranges_partition_exactly():
    from cloud_map_evaluation_amd.dist import shard_range

    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in cuts) - min(e - b for b, e in cuts) <= 1


def test_single_process_path_needs_no_process_group():
    """world == 1: suite_step must not touch torch.distributed at all."""
    import oracle
    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd import synth
    from cloud_map_evaluation_amd.engine import Param

    est, gt = synth.cube_pair(6000, seed=3)
    est, gt = est.numpy() * 0.4, gt.numpy() * 0.4
    eng = OracleShardEngine()
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=0.5)
    r = medist.suite_step(eng, None, None, est, gt, P)
    o = oracle.reg_stats(est, gt, 1.0, 0, TRUNC)
    assert np.array_equal(r["est_gt"]["number"], o.number)
    np.testing.assert_allclose(r["ac"], o.rmse, rtol=1e-12)
    np.testing.assert_allclose(r["com"], o.fitness, rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# spatial slab mode (suite_step_slab): the same protocol the multi-GPU run uses, with a CPU stand-in engine
# ---------------------------------------------------------------------------------------------------------------
class OracleSlabEngine(OracleShardEngine):
    """Stand-in for Engine in slab mode: holds only slab + halo of each cloud, owns the slab."""

    def __init__(self):
        super().__init__()
        self.slab = None
        self.owned = {}
        self.unres = {}

    def set_slab(self, axis, lo=0.0, hi=0.0, halo=0.0):
        self.slab = None if axis < 0 else (axis, lo, hi, lo - halo, hi + halo)

    def upload(self, slot, xyz, T=None, cell_size=0.0):
        import oracle

        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        if T is not None:
            xyz = oracle.transform(xyz, T)
        if self.slab is None:
            self.cloud[slot], self.owned[slot] = xyz, np.ones(len(xyz), bool)
            return
        a, lo, hi, rlo, rhi = self.slab
        v = xyz[:, a]
        keep = (v >= rlo) & (v < rhi)
        self.cloud[slot] = xyz[keep]
        self.owned[slot] = (v[keep] >= lo) & (v[keep] < hi)

    def mme(self, slot, radius, min_k, per_point=False):
        import oracle

        if len(self.cloud[slot]) == 0:
            return 0.0, None, None, 0, 0.0
        _, ent, val, _, _ = oracle.mme(self.cloud[slot], radius, min_k)
        o = self.owned[slot]
        return 0.0, None, None, int(val[o].sum()), float(ent[o].sum())

    def nn1(self, q, r, fetch=False):
        import oracle

        nq = len(self.cloud[q])
        d2 = oracle.nn1(self.cloud[r], self.cloud[q])[1] if len(self.cloud[r]) and nq else np.full(nq, np.inf)
        self.d2[q] = d2
        if self.slab is None:
            self.unres[q] = np.zeros(0, np.int64)
        else:
            a, lo, hi, rlo, rhi = self.slab
            v = self.cloud[q][:, a]
            face = np.minimum(v - rlo, rhi - v)
            self.unres[q] = np.nonzero(self.owned[q] & ~(d2 < face * face))[0]
        return None, None

    def nn_unresolved_count(self, q):
        return len(self.unres[q])

    def nn_unresolved(self, q, with_d2=False):
        import torch

        xyz = self.cloud[q][self.unres[q]].copy()
        if with_d2:
            xyz = np.concatenate([xyz, self.d2[q][self.unres[q]][:, None]], 1)
        return torch.from_numpy(xyz)

    def nn_points(self, r, xyz, bound=None, covered=None, axis=0):  # (covered: a search hint of the real engine, the answer is the same)
        import oracle
        import torch

        pts = np.ascontiguousarray(xyz.numpy())
        if len(self.cloud[r]) == 0 or len(pts) == 0:
            d2 = np.full(len(pts), np.inf)
        else:
            d2 = oracle.nn1(self.cloud[r], pts)[1]
        if bound is not None:
            d2 = np.minimum(d2, bound.numpy())
        return torch.from_numpy(d2)

    def nn_patch(self, q, d2):
        self.d2[q][self.unres[q]] = np.minimum(self.d2[q][self.unres[q]], d2.numpy())

    def _slab(self, n):  # statistics run over the owned points only
        raise NotImplementedError

    def nn_partial_sums(self, q, gate, mode, trunc):
        d2 = self.d2[q][self.owned[q]]
        keep = self._gate(d2, gate, mode)
        d = np.sqrt(d2)
        out = types.SimpleNamespace(n_query=len(d2), n_corr=int(keep.sum()), n_inl=[], sum_d=[], sum_d2=[],
                                    sum_sqrt_all=float(d.sum()))
        for t in trunc:
            inl = keep & (d <= t)
            out.n_inl.append(int(inl.sum()))
            out.sum_d.append(float(d[inl].sum()))
            out.sum_d2.append(float(d2[inl].sum()))
        return out

    def nn_sigma_sums(self, q, gate, mode, mean):
        d2 = self.d2[q][self.owned[q]]
        d = np.sqrt(d2[self._gate(d2, gate, mode)])
        return np.array([((d - m) ** 2).sum() for m in mean])

    def voxel_partials(self, slot, vs):
        p = self.cloud[slot][self.owned[slot]]
        if len(p) == 0:
            return np.zeros((0, 3), np.int32), np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros((0, 3, 3))
        keys = np.floor(p / vs).astype(np.int32)
        uk, inv = np.unique(keys, axis=0, return_inverse=True)
        inv = inv.ravel()
        n = np.bincount(inv, minlength=len(uk)).astype(np.int32)
        mu = np.zeros((len(uk), 3))
        np.add.at(mu, inv, p)
        mu /= n[:, None]
        c = p - mu[inv]
        m2 = np.zeros((len(uk), 3, 3))
        np.add.at(m2, inv, c[:, :, None] * c[:, None, :])
        return uk, n, mu, m2

    def w2_batch(self, mu1, s1, n1, mu2, s2, n2):
        import oracle

        return np.array([oracle.w2_gaussian(mu1[i], s1[i], int(n1[i]), mu2[i], s2[i], int(n2[i])) for i in range(len(n1))])

    def scs_table(self, keys, w, radius=5):
        import oracle

        return oracle.scs(keys, w, radius)


def _slab_worker(rank, world, port, est, gt, T, q):
    import torch
    import torch.distributed as dist

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Param

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=0.5, initial_matrix_=T)
        res = medist.suite_step_slab(OracleSlabEngine(), dist, torch.device("cpu"), est, gt, P, rank, world, halo=0.3)
        q.put((rank, {k: (v if not isinstance(v, dict) else {kk: np.asarray(vv) for kk, vv in v.items()}) for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_slab_suite_gloo_equals_single_process_oracle(world):
    import torch.multiprocessing as mp

    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(12000, seed=33)
    est, gt = est.numpy() * 0.5, gt.numpy()[:11000] * 0.5
    est = np.concatenate([est, est[:60] + np.array([0.9, 0.0, 1.5])])  # their nearest GT point is in another slab
    T = np.eye(4)
    T[:3, 3] = [0.004, -0.003, 0.002]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, est, gt, T, q)) for r in range(world)]
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
    o_v = oracle.awd_scs(oracle.VoxelMap(gt, 0.5), oracle.VoxelMap(est_t, 0.5))
    for rank in range(world):
        r = results[rank]
        assert r["n_cross_rank_queries"] > 0  # the protocol's cross-rank step was exercised
        for got, exp in ((r["est_gt"], o_eg), (r["gt_est"], o_ge)):
            assert got["n_corr"] == exp.n_corr
            assert np.array_equal(got["number"], exp.number)
            assert np.array_equal(got["fitness"], exp.fitness)
            for k in ("mean", "rmse", "sigma"):
                np.testing.assert_allclose(got[k], getattr(exp, k), rtol=1e-12)
        np.testing.assert_allclose(r["cd"], oracle.chamfer(est_t, gt), rtol=1e-12)
        assert r["mme_valid"] == o_me[3]
        np.testing.assert_allclose(r["mme_est"], o_me[0], rtol=1e-12)
        np.testing.assert_allclose(r["mme_gt"], o_mg[0], rtol=1e-12)
        assert r["n_w"] == len(o_v["rows"])
        np.testing.assert_allclose(r["awd"], o_v["awd"], rtol=1e-9)
        np.testing.assert_allclose(r["scs"], o_v["scs"], rtol=1e-9)


def test_merge_voxel_partials_is_chan_exact():
    """Splitting a voxel's points over ranks and merging the partials reproduces the whole-voxel Gaussian."""
    from cloud_map_evaluation_amd.dist import merge_voxel_partials

    rng = np.random.default_rng(0)
    p = rng.normal(0, 0.3, (5000, 3)) + np.array([7.0, -3.0, 1.0])
    e = OracleSlabEngine()
    rows = []
    for part in np.array_split(p, 4):
        e.cloud[0], e.owned[0] = part, np.ones(len(part), bool)
        k, n, mu, m2 = e.voxel_partials(0, 1.0)
        rows.append(np.concatenate([k.astype(float), n[:, None].astype(float), mu, m2.reshape(-1, 9)], 1))
    keys, n, mu, sig = merge_voxel_partials(np.concatenate(rows))
    e.cloud[0], e.owned[0] = p, np.ones(len(p), bool)
    k2, n2, mu2, m22 = e.voxel_partials(0, 1.0)
    assert np.array_equal(keys, k2) and np.array_equal(n, n2)
    np.testing.assert_allclose(mu, mu2, rtol=1e-13)
    exp = m22.copy()
    big = n2 > 10
    exp[big] = exp[big] / (n2[big] - 1.0)[:, None, None] / (n2[big] - 1.0)[:, None, None]
    np.testing.assert_allclose(sig, exp, rtol=1e-10, atol=1e-18)


class _TwoLaneEngine(OracleShardEngine):
    """The stand-in with a twin(): the second lane of an overlapped step then runs on a real second thread."""

    def __init__(self, fail_in=None):
        super().__init__()
        self.fail_in = fail_in
        self.calls = []

    def twin(self):
        return self

    def voxel_build(self, slot, vs):
        self.calls.append(("voxel_build", slot))
        if self.fail_in == "voxel_build":
            raise RuntimeError("lane failure")

    def nn1(self, q, r, fetch=False):
        import threading

        self.calls.append(("nn1", q, threading.current_thread() is threading.main_thread()))
        return super().nn1(q, r, fetch)


def _small_pair():
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(6000, seed=4)
    return est.numpy(), gt.numpy()


def test_overlapped_driver_equals_sequential_driver_on_cpu():
    """dist._Lane on CPU: same scalars with and without the second lane; the ground-truth -> map search runs on the lane's
    thread, the map -> ground-truth one on the caller's."""
    import torch

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Param

    est, gt = _small_pair()
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=2.0)
    seq = medist.suite_step(_TwoLaneEngine(), None, torch.device("cpu"), est, gt, P, overlap=False)
    eng = _TwoLaneEngine()
    ovl = medist.suite_step(eng, None, torch.device("cpu"), est, gt, P, overlap=True)
    assert seq["n_w"] > 0 and np.isfinite(seq["awd"])
    for k in ("cd", "mme_est", "mme_gt", "awd", "scs", "mme_valid", "n_w"):
        assert seq[k] == ovl[k], k
    for d in ("est_gt", "gt_est"):
        for k in ("mean", "rmse", "sigma", "number", "fitness"):
            assert np.array_equal(np.asarray(seq[d][k]), np.asarray(ovl[d][k])), (d, k)
    on_main = {c[1]: c[2] for c in eng.calls if c[0] == "nn1"}
    assert on_main == {0: True, 1: False}
    assert [c for c in eng.calls if c[0] == "voxel_build"] == [("voxel_build", 1), ("voxel_build", 0)]


@pytest.mark.timeout(120)
def test_a_failure_on_the_second_lane_surfaces_and_does_not_hang():
    import torch

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Param

    est, gt = _small_pair()
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=0.5)
    with pytest.raises(RuntimeError, match="lane failure"):
        medist.suite_step(_TwoLaneEngine(fail_in="voxel_build"), None, torch.device("cpu"), est, gt, P, overlap=True)


# ---------------------------------------------------------------------------------------------------------------
# distributed input (suite_step_dist): every rank starts with 1/world of each cloud, one all-to-all halo exchange
# ---------------------------------------------------------------------------------------------------------------
def numpy_lattice_histograms(p, e0):
    """me_lattice_histograms_device restated in numpy (csrc/me_dist.hip: k_lattice_range + k_lattice_hist)."""
    import torch
    from cloud_map_evaluation_amd.dist import LATTICE_BINS as B

    fin = np.isfinite(p)
    ninf = np.array([int(np.sum(np.isneginf(p[:, a]))) for a in range(3)], dtype=np.int64)
    level, origin = 0, np.zeros(3, np.int64)
    hist = np.zeros((3, B), np.int32)
    if fin.any():
        b0 = [np.floor(p[fin[:, a], a] * 2.0 ** -e0).astype(np.int64) for a in range(3)]
        while any(len(b) and (b.max() >> level) - (b.min() >> level) + 1 > B for b in b0):
            level += 1
        for a in range(3):
            if len(b0[a]):
                origin[a] = b0[a].min() >> level
                hist[a] = np.bincount((b0[a] >> level) - origin[a], minlength=B)
    return level, origin, ninf, torch.from_numpy(hist)


class OracleDistEngine(OracleSlabEngine):
    """Stand-in for Engine with the distributed-input primitives restated in numpy (host tensors, gloo collectives)."""

    def __init__(self):
        super().__init__()
        self.merged = {}

    def transform_points(self, xyz, T):
        import oracle
        import torch

        return torch.from_numpy(oracle.transform(xyz.numpy(), T))

    def halo_pack(self, xyz, axis, cuts, halo):
        import torch

        p = xyz.numpy()
        v = p[:, axis]
        segs, counts = [], []
        for k in range(len(cuts) - 1):
            keep = (v >= cuts[k] - halo) & (v < cuts[k + 1] + halo)  # the filter of me_set_slab, input order kept
            segs.append(p[keep])
            counts.append(int(keep.sum()))
        return torch.from_numpy(np.concatenate(segs) if segs else np.zeros((0, 3))), counts

    def lattice_histograms(self, xyz, e0):
        return numpy_lattice_histograms(xyz.numpy(), e0)

    def upload(self, slot, xyz, T=None, cell_size=0.0):
        super().upload(slot, xyz.numpy() if hasattr(xyz, "numpy") else xyz, T, cell_size)

    def voxel_partial_rows(self, slot, vs):
        import torch

        k, n, mu, m2 = self.voxel_partials(slot, vs)
        if len(n) == 0:
            return torch.zeros((0, 16), dtype=torch.float64)
        return torch.from_numpy(np.concatenate([k.astype(np.float64), n[:, None].astype(np.float64), mu, m2.reshape(-1, 9)], 1))

    def voxel_merge(self, slot, vs, rows):
        from cloud_map_evaluation_amd.dist import merge_voxel_partials

        self.merged[slot] = merge_voxel_partials(rows.numpy())

    def calculateVMD(self, vs, rows=False):
        from cloud_map_evaluation_amd.dist import awd_scs_from_tables

        return awd_scs_from_tables(self, self.merged[0], self.merged[1])


def _dist_worker(rank, world, port, est, gt, T, q):
    import torch
    import torch.distributed as dist

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd.engine import Param

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if os.environ.get("ME_TEST_CROSS_CAP"):  # force the overflow path of the folded cross-rank all-gather
        medist._CROSS_CAP = int(os.environ["ME_TEST_CROSS_CAP"])
    if os.environ.get("ME_TEST_VOX_CAP"):  # ... and of the folded statistics gather
        medist._VOX_CAP = int(os.environ["ME_TEST_VOX_CAP"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = Param(icp_max_distance_=1.0, nn_radius_=0.1, trunc_dist_=TRUNC, vmd_voxel_size_=0.5, initial_matrix_=T)
        # this rank's part of the input: a contiguous piece of each cloud (rank 1 of a 3-rank job holds NO ground truth at all)
        be, ee = medist.shard_range(len(est), rank, world)
        if world == 3:
            bg, eg = [(0, len(gt) // 2), (0, 0), (len(gt) // 2, len(gt))][rank]
        else:
            bg, eg = medist.shard_range(len(gt), rank, world)
        res = medist.suite_step_dist(OracleDistEngine(), dist, torch.device("cpu"), torch.from_numpy(est[be:ee].copy()),
                                     torch.from_numpy(gt[bg:eg].copy()), P, rank, world, halo=0.3, overlap=False)
        q.put((rank, {k: (v if not isinstance(v, dict) else {kk: np.asarray(vv) for kk, vv in v.items()}) for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,cap,protocol", [(2, 0, "lean"), (3, 0, "lean"), (4, 0, "lean"), (2, 5, "lean"), (3, 0, "lean_vox_overflow"), (2, 0, "classic"), (3, 5, "classic")])
def test_distributed_input_suite_gloo_equals_single_process_oracle(world, cap, protocol, monkeypatch):
    """1/world of each cloud per rank -> lean: cuts, halo and every message size from ONE gather of lattice histograms (classic,
    ME_DIST_LEAN=0: a gathered sample + a count all-to-all) -> all-to-all halo exchange -> local passes -> batched cross-rank resolve
    (one fixed-capacity all-gather with the counts in-band; cap = 5 forces its overflow fallback) -> partial sums and voxel rows in one
    fixed-capacity gather (lean_vox_overflow: capacity 3 forces the exact-size gather) -> merged voxel tables: every rank ends with
    the single-process oracle's answer."""
    import torch.multiprocessing as mp

    if cap:
        monkeypatch.setenv("ME_TEST_CROSS_CAP", str(cap))
    if protocol == "classic":
        monkeypatch.setenv("ME_DIST_LEAN", "0")
    if protocol == "lean_vox_overflow":
        monkeypatch.setenv("ME_TEST_VOX_CAP", "3")

    import oracle
    from cloud_map_evaluation_amd import synth

    est, gt = synth.cube_pair(12000, seed=35)
    est, gt = est.numpy() * 0.5, gt.numpy()[:11000] * 0.5
    est = np.concatenate([est, est[:60] + np.array([0.9, 0.0, 1.5])])  # their nearest GT point is in another slab
    rng = np.random.default_rng(1)
    est, gt = est[rng.permutation(len(est))], gt[rng.permutation(len(gt))]  # a rank's piece of the file is NOT spatially compact
    T = np.eye(4)
    T[:3, 3] = [0.004, -0.003, 0.002]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, est, gt, T, q)) for r in range(world)]
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
    o_v = oracle.awd_scs(oracle.VoxelMap(gt, 0.5), oracle.VoxelMap(est_t, 0.5))
    for rank in range(world):
        r = results[rank]
        assert r["n_est"] == len(est) and r["n_gt"] == len(gt)
        assert r["n_cross_rank_queries"] > 0  # the protocol's cross-rank step was exercised
        for got, exp in ((r["est_gt"], o_eg), (r["gt_est"], o_ge)):
            assert got["n_corr"] == exp.n_corr
            assert np.array_equal(got["number"], exp.number)
            assert np.array_equal(got["fitness"], exp.fitness)
            for k in ("mean", "rmse", "sigma"):
                np.testing.assert_allclose(got[k], getattr(exp, k), rtol=1e-12)
        np.testing.assert_allclose(r["cd"], oracle.chamfer(est_t, gt), rtol=1e-12)
        assert r["mme_valid"] == o_me[3]
        np.testing.assert_allclose(r["mme_est"], o_me[0], rtol=1e-12)
        np.testing.assert_allclose(r["mme_gt"], o_mg[0], rtol=1e-12)
        assert r["n_w"] == len(o_v["rows"])
        np.testing.assert_allclose(r["awd"], o_v["awd"], rtol=1e-9)
        np.testing.assert_allclose(r["scs"], o_v["scs"], rtol=1e-9)


def test_dist_slab_cuts_single_process_are_balanced_and_ascending():
    import torch

    from cloud_map_evaluation_amd import dist as medist
    from cloud_map_evaluation_amd import synth

    _, gt = synth.cube_pair(50_000, seed=2)
    axis, cuts = medist.dist_slab_cuts(gt, None, torch.device("cpu"), 4)
    assert cuts[0] == -np.inf and cuts[-1] == np.inf and all(cuts[i] < cuts[i + 1] for i in range(4))
    v = gt[:, axis].numpy()
    share = [((v >= cuts[k]) & (v < cuts[k + 1])).mean() for k in range(4)]
    assert abs(sum(share) - 1.0) < 1e-12 and max(share) < 0.33  # equal-count slabs (cube faces make the histogram lumpy)
    assert medist.dist_slab_cuts(gt[:0], None, torch.device("cpu"), 1)[1] == [-np.inf, np.inf]
    same = torch.ones((100, 3), dtype=torch.float64)
    _, c = medist.dist_slab_cuts(same, None, torch.device("cpu"), 3)
    assert all(c[i] < c[i + 1] for i in range(3))


def _plan_for(parts_by_rank, halo):
    """lattice_plan on numpy parts [[est_r, gt_r] per rank] -> (axis, cuts, halo_eff, counts, totals)."""
    import torch

    from cloud_map_evaluation_amd import dist as medist

    e0 = medist.lattice_e0(halo)
    eng = OracleDistEngine()
    msgs = [medist.lattice_message(eng, [torch.from_numpy(np.ascontiguousarray(c)) for c in parts], e0) for parts in parts_by_rank]
    return medist.lattice_plan(torch.stack(msgs), len(parts_by_rank), halo, e0)


def _membership_counts(parts_by_rank, axis, cuts, halo):
    eng = OracleDistEngine()
    import torch

    return np.array([[eng.halo_pack(torch.from_numpy(np.ascontiguousarray(c)), axis, cuts, halo)[1] for c in parts] for parts in parts_by_rank])


@pytest.mark.parametrize("case", ["uniform", "negative_and_edges", "separated_parts", "huge_extent", "nonfinite_and_empty", "degenerate"])
def test_lattice_plan_predicts_every_message_of_the_halo_exchange(case):
    """The lean exchange: cuts at bin edges of an absolute power-of-two lattice, a halo of whole bins, and the per-destination counts
    of every source rank computed from the gathered histograms alone — equal, number for number, to what halo_pack's comparisons
    (me_set_slab's filter) select with those cuts."""
    rng = np.random.default_rng(7)
    world, halo = 4, 0.3
    if case == "uniform":
        parts = [[rng.uniform(-3, 9, (4000, 3)) * [1, 0.3, 0.1], rng.uniform(-3, 9, (3500, 3)) * [1, 0.3, 0.1]] for _ in range(world)]
    elif case == "negative_and_edges":
        w = 2.0 ** -6  # (a multiple of every bin width up to 2^-6: coordinates ON bin edges)
        parts = [[np.round(rng.uniform(-20, -5, (3000, 3)) / w) * w, np.round(rng.uniform(-20, -5, (2500, 3)) / w) * w] for _ in range(world)]
    elif case == "separated_parts":  # every rank's piece is compact and far from the others: the combined window is coarsened
        parts = [[rng.uniform(0, 60, (2000, 3)) + [400.0 * r, 0, 0], rng.uniform(0, 60, (2000, 3)) + [400.0 * r, 0, 0]] for r in range(world)]
    elif case == "huge_extent":
        parts = [[rng.uniform(-4000, 9000, (3000, 3)), rng.uniform(-4000, 9000, (3000, 3))] for _ in range(world)]
    elif case == "nonfinite_and_empty":
        a = rng.uniform(0, 10, (2000, 3))
        a[5, 0], a[6, 0], a[7, 0], a[8, 1] = -np.inf, np.inf, np.nan, -np.inf
        parts = [[a, rng.uniform(0, 10, (1500, 3))], [np.zeros((0, 3)), rng.uniform(0, 10, (900, 3))], [rng.uniform(0, 10, (10, 3)), np.zeros((0, 3))],
                 [np.zeros((0, 3)), np.zeros((0, 3))]]
    else:  # all points in one bin of the slab axis
        parts = [[np.tile([[1.0, 2.0, 3.0]], (500, 1)) + [0, 1e-3 * r, 0], np.tile([[1.0, 2.0, 3.0]], (400, 1))] for r in range(world)]
    axis, cuts, halo_eff, counts, totals = _plan_for(parts, halo)
    assert len(cuts) == world + 1 and cuts[0] == -np.inf and cuts[-1] == np.inf
    assert all(cuts[k] < cuts[k + 1] for k in range(world))
    assert halo_eff >= halo
    if case in ("uniform", "negative_and_edges", "nonfinite_and_empty", "degenerate"):
        assert halo_eff <= halo * 17 / 16 + 1e-12  # the finest lattice: whole bins of at most halo / 16
    assert totals == [sum(len(p[c]) for p in parts) for c in range(2)]
    assert np.array_equal(counts, _membership_counts(parts, axis, cuts, halo_eff))
    if case == "uniform":
        assert axis == 0
        owned = np.array([sum(int(((p[c][:, axis] >= cuts[k]) & (p[c][:, axis] < cuts[k + 1])).sum()) for p in parts for c in range(2)) for k in range(world)])
        assert owned.max() - owned.min() <= 0.02 * owned.sum()  # equal counts of both clouds together, to a bin's worth


def test_statistics_gather_capacity_follows_the_world_size():
    from cloud_map_evaluation_amd import dist as medist

    assert medist._VOX_CAP is None  # (tests override it through ME_TEST_VOX_CAP in their worker processes only)
    assert [medist._vox_cap(w) for w in (1, 2, 4, 8, 16, 64)] == [8192, 4096, 2048, 1024, 512, 512]
    assert medist._FOLD_HEAD * 16 >= medist.VEC_LEN + 6  # the header rows hold the partial sums and the six counts
