#!/usr/bin/env python
"""The C++ drop-in (`cloud_map_evaluation_amd/host/map_eval`) on the bench pair: writes the c4_multisession clouds as binary PCD
files (fp64), runs the binary on the reference's YAML format (evaluate_using_initial: true, no down-sampling: the metric phase is
ONE me_run_suite_from call from process()'s single thread) and keeps the line it prints about that call — the host binary's wall
time for the metric phase, next to bench.py's step time on the same box.  Run on the GPU box from the repo root:
    python profiles/host_one_call.py [points] > gpurun_out/host_one_call.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_pcd(path, pts):
    import numpy as np

    n = len(pts)
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\n"
           f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(np.ascontiguousarray(pts, dtype="<f8").tobytes())


def main():
    import torch

    from cloud_map_evaluation_amd import synth

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    est, gt = synth.multisession_pair(n, 3, density=2500.0, seed=100, device=torch.device("cuda", 0))
    est, gt = est.cpu().numpy(), gt.cpu().numpy()
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "cloud_map_evaluation_amd", "host", "map_eval")
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        os.mkdir(os.path.join(d, "est"))
        write_pcd(os.path.join(d, "est", "map.pcd"), est)
        write_pcd(os.path.join(d, "gt.pcd"), gt)
        del est, gt
        cfg = f"""registration_methods: 2
icp_max_distance: 1.0
accuracy_level: [0.2, 0.1, 0.08, 0.05, 0.01]
initial_matrix:
  - [1.0, 0.0, 0.0, 0.0]
  - [0.0, 1.0, 0.0, 0.0]
  - [0.0, 0.0, 1.0, 0.0]
  - [0.0, 0.0, 0.0, 1.0]
estimate_map_path: {d}/est/
pcd_file_name: map.pcd
gt_map_path: {d}/gt.pcd
scene_name: c4_multisession
save_immediate_result: false
evaluate_mme: true
use_tbb_mme: true
evaluate_gt_mme: true
nn_radius: 0.1
evaluate_using_initial: true
evaluate_noise_gt: false
vmd_voxel_size: 3.0
downsample_size: 0.0
use_visualization: false
enable_debug: true
"""
        open(os.path.join(d, "config.yaml"), "w").write(cfg)
        # ONE process, the evaluation three times (`--repeat 3`): run 1 is the cold start every fresh process pays (runtime + code
        # objects, every device allocation, GPU clocks coming up from idle after ~1 s of file reading), runs 2-3 the same call warm
        t0 = time.perf_counter()
        r = subprocess.run([exe, os.path.join(d, "config.yaml"), "--repeat", "3"], capture_output=True, text=True, timeout=1500)
        wall = time.perf_counter() - t0
        runs = [{"metric_phase_ms": float(m.group(1)), "stages": m.group(4)} for m in re.finditer(
            r"INFO: metric phase, one me_run_suite_from call \(two lanes\): ([0-9.]+) ms for (\d+) \+ (\d+) points \[(.*)\]", r.stdout)]
        proc = {"rc": r.returncode, "process_wall_s_for_3_evaluations": wall, "err": r.stderr[-300:] if r.returncode else None,
                "tail": r.stdout[-600:] if not runs else None}
        res = open(os.path.join(d, "est", "map_results", "map_results.txt")).read().splitlines()[-12:]
    warm = min((x["metric_phase_ms"] for x in runs[1:]), default=None)
    print(json.dumps({"what": "host/map_eval on the c4_multisession pair from PCD files (host memory -> results): the metric phase of "
                              "MapEval::process() is one me_run_suite_from call (ME_SUITE_OVERLAP | ME_SUITE_PIN_HOST_INPUT) made from its "
                              "single thread; H2D included; run 1 = cold process, runs 2-3 = the same call warm (--repeat 3)",
                      "points": n, "process": proc, "runs": runs, "metric_phase_ms_cold": runs[0]["metric_phase_ms"] if runs else None,
                      "metric_phase_ms_warm": warm, "mpts_per_s_warm": (2 * n / 1e6 / (warm / 1e3)) if warm else None,
                      "map_results_tail": res}, indent=1))


if __name__ == "__main__":
    main()
