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
