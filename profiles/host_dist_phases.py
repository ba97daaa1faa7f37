"""Wall-clock split of the C++ host (`map_eval`, host/map_eval_dist.cpp) on a campus pair written as binary PCD files:
python profiles/host_dist_phases.py [points] -> JSON.  Three runs of the same config: single GPU; ONE rank with every collective
through RCCL (MAPEVAL_FORCE_DIST=1); two ranks sharing the GPU over the file transport (tests' stand-in for a second device: its
collectives are files, so its phases are an upper bound).  evaluate_using_initial: true, save_immediate_result: false."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cloud_map_evaluation_amd import synth  # noqa: E402

EXE = os.path.join(ROOT, "cloud_map_evaluation_amd", "host", "map_eval")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000


def write_pcd(path, pts):
    m = len(pts)
    hdr = (f"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH {m}\nHEIGHT 1\n"
           f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {m}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(np.ascontiguousarray(pts, dtype="<f8").tobytes())


est, gt = synth.campus_pair(n, density=2500.0, seed=5)
est, gt = est.numpy(), gt.numpy()
out = {"points": [int(len(est)), int(len(gt))]}
with tempfile.TemporaryDirectory() as td:
    os.makedirs(os.path.join(td, "est"))
    write_pcd(os.path.join(td, "est", "map.pcd"), est)
    write_pcd(os.path.join(td, "gt.pcd"), gt)
    for name, num_gpus, env in (("single_gpu", 1, {}), ("one_rank_rccl", 1, {"MAPEVAL_FORCE_DIST": "1"}),
                                ("two_ranks_one_gpu_file_transport", 2, {"MAPEVAL_COMM": "file", "MAPEVAL_SINGLE_DEVICE": "1"})):
        res = os.path.join(td, "res_" + name)
        os.makedirs(res)
        cfg = os.path.join(td, name + ".yaml")
        open(cfg, "w").write(f"""registration_methods: 2
icp_max_distance: 1.0
accuracy_level: [0.2, 0.1, 0.08, 0.05, 0.01]
initial_matrix:
  - [1.0, 0.0, 0.0, 0.0]
  - [0.0, 1.0, 0.0, 0.0]
  - [0.0, 0.0, 1.0, 0.0]
  - [0.0, 0.0, 0.0, 1.0]
estimate_map_path: {td}/est/
gt_map_path: {td}/gt.pcd
scene_name: campus
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
enable_debug: false
num_gpus: {num_gpus}
""")
        e = dict(os.environ)
        e.update(env)
        t0 = time.time()
        r = subprocess.run([EXE, cfg], capture_output=True, text=True, env=e, timeout=600)
        rec = {"wall_s": round(time.time() - t0, 3), "rc": r.returncode}
        m = re.search(r"multi-GPU phases on rank 0 \[ms\]:(.*)", r.stdout)
        if m:
            rec["phases_ms"] = {k: float(v) for k, v in re.findall(r"(\w+)=([\d.]+)", m.group(1))}
        try:
            txt = open(os.path.join(td, "est", "map_results", "map_results.txt")).read()
            for key in ("RMSE/AC", "FULL CD", "MME", "VMD"):
                mm = re.findall(rf"^{re.escape(key)}: (.*)$", txt, flags=re.M)
                if mm:
                    rec[key] = mm[-1].strip()
        except OSError:
            pass
        tm = re.findall(r"^(?:Time|time)[^\n]*$", r.stdout, flags=re.M)
        if tm:
            rec["time_lines"] = tm[-3:]
        if r.returncode != 0:
            rec["stderr_tail"] = r.stderr[-400:]
        out[name] = rec
print(json.dumps(out, indent=1))
