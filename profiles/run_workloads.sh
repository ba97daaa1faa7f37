#!/bin/bash
# The bench line of `python bench.py` and of the other --workloads, on the GPU box from the repo root:
#   bash profiles/run_workloads.sh r02      -> gpurun_out/summary_<tag>/{bench_default.json, workloads.json}
TAG=${1:-r03}
OUT=gpurun_out/summary_$TAG
mkdir -p "$OUT"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT" <<'PY'
import json, subprocess, sys
out = {}
for w in ("campus", "c3_20m", "c5_tunnel"):
    r = subprocess.run([sys.executable, "bench.py", "--workload", w, "--cpu-baseline", "off", "--steps", "3"], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out[w] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
              "n_est": d["config"]["n_est"], "n_gt": d["config"]["n_gt"], "vmd_voxel_size": d["config"]["vmd_voxel_size"],
              "h2d_inclusive_ms_per_step": (d.get("h2d_inclusive") or {}).get("ms_per_step"),
              "kernel_ms_per_step": d["roofline"]["kernel_ms_per_step"], "results": d["results"]}
out["_note"] = "python bench.py --workload <w> --cpu-baseline off --steps 3 (profiles/run_workloads.sh)"
json.dump(out, open(sys.argv[1] + "/workloads.json", "w"), indent=1)
PY
tail -c 600 "$OUT/bench_default.json"
