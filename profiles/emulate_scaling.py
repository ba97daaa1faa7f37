#!/usr/bin/env python
"""Strong-scaling estimate on ONE GPU: runs the slab-mode step of every rank of an N-rank job one after another with
stub collectives (all_reduce = identity, all_gather = own buffer repeated), so the metric VALUES are partial but the
per-rank compute time is what a real rank would spend between collectives.  usage: python profiles/emulate_scaling.py"""
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

    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.rank

    def all_reduce(self, t, op=None):
        return None

    def all_gather(self, parts, buf):
        for p in parts:
            p.copy_(buf)

    def barrier(self):
        pass


def main(points=50_000_000, overlap=True):
    dev = torch.device("cuda", 0)
    est, gt = synth.campus_pair(points, density=2500.0, seed=100, device=dev)
    P = Param(icp_max_distance_=1.0, nn_radius_=0.1, vmd_voxel_size_=3.0)
    eng = Engine(0)
    out = {}
    for world in (1, 2, 4, 8):
        per_rank = []
        for rank in range(world):
            fd = FakeDist(world, rank)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if world == 1:
                    medist.suite_step(eng, None, dev, est, gt, P, True, overlap=overlap)
                else:
                    medist.suite_step_slab(eng, fd, dev, est, gt, P, rank, world, True, halo=1.0, overlap=overlap)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            per_rank.append(best * 1e3)
        out[world] = {"max_ms": max(per_rank), "mean_ms": sum(per_rank) / len(per_rank)}
        print(world, out[world], flush=True)
    base = out[1]["max_ms"]
    print(json.dumps({"points": points, "overlap": overlap, "per_world": out,
                      "speedup_vs_1": {w: base / v["max_ms"] for w, v in out.items()}}))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000, overlap=("--no-overlap" not in sys.argv))
