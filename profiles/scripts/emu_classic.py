#!/usr/bin/env python
"""Strong-scaling ESTIMATE on ONE GPU for the distributed-input step (dist.suite_step_dist): every rank of an N-rank job is
run one after another with a stand-in process group whose collectives return what the real ones would (the all-to-all
delivers the points the other ranks would have sent — precomputed, untimed; all-reduces are identities; all-gathers deliver
what the other ranks DID contribute in an untimed recording pass of every rank's step (round 4 — until then they repeated the
rank's own message, which made every rank search its own open queries world - 1 more times in the cross-rank step: queries next
to its own slab, the ones its tree cannot prune, instead of the other ranks' which it mostly prunes at the root), so a rank's
time is the compute + host work it would spend between collectives.  RCCL latency of
the ~10 small collectives (~0.3-0.5 ms per step) and the halo payload (2 x 24 B x N / world^2 per link: < 0.3 ms at 8 ranks
over xGMI) are NOT included.  usage: python profiles/emulate_scaling.py [points] [--workload campus|c4_multisession] [--worlds 1,2,4,8]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from cloud_map_evaluation_amd import dist as medist, synth  # noqa: E402
from cloud_map_evaluation_amd.engine import Engine, Param  # noqa: E402


class FakeDist:
    class ReduceOp:
        SUM, MAX, MIN = "sum", "max", "min"

    def __init__(self, world, rank, recv_counts=None, recv_points=None, record=None, replay=None):
        self.world, self.rank = world, rank
        self.recv_counts, self.recv_points = recv_counts, recv_points
        self.record, self.replay = record, replay  # {gather index within the step: {rank: message}}
        self.n_gather = 0

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.rank

    def all_reduce(self, t, op=None):
        return None

    def all_gather(self, parts, buf):
        idx = self.n_gather
        self.n_gather += 1
        if self.record is not None:
            self.record.setdefault(idx, {})[self.rank] = buf.clone()
        for k, p in enumerate(parts):
            src = buf
            if self.replay is not None and k != self.rank:
                r = self.replay.get(idx, {}).get(k)
                if r is not None and r.shape == buf.shape and r.dtype == buf.dtype:
                    src = r  # (a message of another shape — a rank whose step took another branch — falls back to the own copy)
            p.copy_(src)

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        if out.dtype == torch.int64:
            out.copy_(self.recv_counts)   # what every rank sends to this one
        else:
            out.copy_(self.recv_points)

    def barrier(self):
        pass


def main(points, workload, worlds=(1, 2, 4, 8)):
    dev = torch.device("cuda", 0)
    if workload == "c4_multisession":
        est, gt = synth.multisession_pair(points, 3, density=2500.0, seed=100, device=dev)
    else:
        est, gt = synth.scan_pair(points, density=2500.0, seed=100, device=dev)
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=3.0)
    eng = Engine(0)
    halo = 1.0
    out = {}
    for world in worlds:
        pieces = [(est[slice(*medist.shard_range(est.shape[0], r, world))], gt[slice(*medist.shard_range(gt.shape[0], r, world))])
                  for r in range(world)]
        per_rank, detail = [], []
        if world > 1:
            # what the halo exchange delivers to every rank (untimed): the cuts are those every rank computes
            axis, cuts = medist.dist_slab_cuts(gt, None, dev, world, est_part=est)
            packs = [[eng.halo_pack(p, axis, cuts, halo) for p in pc] for pc in pieces]  # [src][cloud] -> (points, counts)
        recorded = {}

        def make_fd(rank, record, replay):
            rc = torch.tensor([[packs[s][c][1][rank] for c in range(2)] for s in range(world)], dtype=torch.int64, device=dev)
            segs = []
            for s in range(world):
                for c in range(2):
                    pts, cnts = packs[s][c]
                    o = sum(cnts[:rank])
                    segs.append(pts[o:o + cnts[rank]])
            return FakeDist(world, rank, rc, torch.cat(segs), record=record, replay=replay)

        if world > 1:
            GLOBAL["cuts"] = (axis, cuts)
            medist.dist_slab_cuts = _patched_cuts
            for rank in range(world):
                fd = make_fd(rank, recorded, None)
                medist.suite_step_dist(eng, fd, dev, pieces[rank][0], pieces[rank][1], P, rank, world, True, halo=halo, overlap=OVERLAP)
            torch.cuda.synchronize()
        for rank in range(world):
            if world == 1:
                fd, args = None, None
            else:
                fd = make_fd(rank, None, recorded)
                GLOBAL["cuts"] = (axis, cuts)
                medist.dist_slab_cuts = _patched_cuts
            best, best_t = 1e9, None
            for rep in range(3):
                eng.timers_enable(rep == 2)
                tw = eng.twin() if OVERLAP else None
                if tw is not None:
                    tw.timers_enable(rep == 2)
                if rep == 2:
                    eng.timers_reset()
                    if tw is not None:
                        tw.timers_reset()
                if fd is not None:
                    fd.n_gather = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if world == 1:
                    res = medist.suite_step(eng, None, dev, est, gt, P, True, overlap=OVERLAP)
                else:
                    res = medist.suite_step_dist(eng, fd, dev, pieces[rank][0], pieces[rank][1], P, rank, world, True, halo=halo, overlap=OVERLAP)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            names = ("mme", "nn_grid", "nn_grid2", "nn1", "nn_far", "nn1_cross", "sort", "morton", "gather", "cells", "voxel", "slab_filter", "halo_pack", "nn_stats")
            best_t = {}
            for e2 in ([eng] + ([eng.twin()] if OVERLAP else [])):  # (both lanes: each context keeps its own timers)
                for k in names:
                    ms, cnt = e2.timer(k)
                    if cnt:
                        best_t[k] = round(best_t.get(k, 0.0) + ms, 2)
                e2.timers_enable(False)
            best_t["open_queries"] = int(res.get("n_cross_rank_queries", 0)) if isinstance(res, dict) else 0
            for k in ("nn1_opened", "nn1_scans", "nn1_max_opened", "nn1_far", "nn1_far_opened", "nn1_far_points", "nn1_far_max"):  # (main lane's walks)
                best_t[k] = int(eng.timer(k)[1])
            medist.dist_slab_cuts = _orig_cuts
            per_rank.append(best * 1e3)
            detail.append(best_t)
        worst = max(range(world), key=lambda r: per_rank[r])
        out[world] = {"max_ms": per_rank[worst], "mean_ms": sum(per_rank) / world, "slowest_rank_kernel_ms": detail[worst],
                      "per_rank_ms": [round(x, 2) for x in per_rank],
                      "per_rank_kernel_ms": detail}
        print(world, out[world], flush=True)
        if world > 1:
            del packs
    base = out[1]["max_ms"] if 1 in out else float("nan")
    print(json.dumps({"points": points, "workload": workload, "driver": "suite_step_dist (distributed input, all-to-all halo)",
                      "per_world": out, "speedup_vs_1": {w: base / v["max_ms"] for w, v in out.items()}}))


_orig_cuts = medist.dist_slab_cuts
GLOBAL = {}
OVERLAP = __import__("os").environ.get("ME_EMU_OVERLAP", "1") != "0"  # 0: one lane (what the phases cost without the other lane)


def _patched_cuts(gt_part, d, cd, w, sample=16384, est_part=None):
    """Collective 1 repeats the rank's own sample here, so a rank alone would cut by its OWN quantiles: do the same device work, then
    hand back the global cuts (those the precomputed exchange was made for)."""
    _orig_cuts(gt_part, None, cd, w, sample, est_part=est_part)
    return GLOBAL["cuts"]


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    wl = "campus"
    if "--workload" in sys.argv:
        wl = sys.argv[sys.argv.index("--workload") + 1]
        a = [x for x in a if x != wl]
    ws = (1, 2, 4, 8)
    if "--worlds" in sys.argv:
        w = sys.argv[sys.argv.index("--worlds") + 1]
        ws = tuple(int(x) for x in w.split(","))
        a = [x for x in a if x != w]
    main(int(a[0]) if a else 50_000_000, wl, ws)
