"""Where the 1-NN fallback's time goes (round 4): python profiles/nn_tail.py [points] [density] — the bench pair, both directions,
kernels timed one at a time, walk counters of k_nn1 and k_nn_far."""
import json
import sys

import torch

sys.path.insert(0, ".")
from cloud_map_evaluation_amd import synth  # noqa: E402
from cloud_map_evaluation_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 2500.0
dev = torch.device("cuda", 0)
est, gt = synth.multisession_pair(n, 3, density=dens, seed=100, device=dev)
out = {"points": n, "density": dens}
with Engine(0) as eng:
    eng.upload(0, est, cell_size=0.1)
    eng.upload(1, gt, cell_size=0.1)
    for name, (a, b) in (("est->gt", (0, 1)), ("gt->est", (1, 0))):
        eng.nn1(a, b, fetch=False)  # warm
        eng.timers_enable(True)
        eng.timers_reset()
        eng.nn1(a, b, fetch=False)
        rec = {}
        for t in ("nn_grid", "nn_grid2", "nn1", "nn_far"):
            ms, cnt = eng.timer(t)
            rec[t + "_ms"] = ms
        for c in ("nn_fallback_queries", "nn1_opened", "nn1_scans", "nn1_points", "nn1_max_opened", "nn1_far", "nn1_far_opened",
                  "nn1_far_points", "nn1_far_max", "nn1_wave_max_10ns", "nn1_wave_sum_10ns", "nn1_waves"):
            rec[c] = eng.timer(c)[1]
        eng.timers_enable(False)
        out[name] = rec
print(json.dumps(out, indent=1))
