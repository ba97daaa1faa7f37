#!/bin/bash
# The bench line of `python bench.py` and of the other --workloads, and (round 4) the DENSITY axis, on the GPU box from the repo root:
#   bash profiles/run_workloads.sh r04      -> gpurun_out/summary_<tag>/{bench_default.json, workloads.json}
# The density sweep also runs the -DME_MME_STATS build (scratch/libmapeval_hip_stats.so:
#   make -C cloud_map_evaluation_amd/csrc EXTRA=-DME_MME_STATS OBJDIR=/tmp/stats_obj OUT=$PWD/scratch/libmapeval_hip_stats.so)
# for the group statistics of the MME kernel: candidates streamed per round, lanes served, accepted pairs per query.
TAG=${1:-r04}
OUT=gpurun_out/summary_$TAG
mkdir -p "$OUT"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT" <<'PY'
import json, os, re, subprocess, sys
out = {}


def bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "bench.py", "--cpu-baseline", "off", "--steps", "3"] + extra, capture_output=True, text=True, env=e)
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def entry(d):
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
            "n_est": d["config"]["n_est"], "n_gt": d["config"]["n_gt"], "vmd_voxel_size": d["config"]["vmd_voxel_size"],
            "h2d_inclusive_ms_per_step": (d.get("h2d_inclusive") or {}).get("ms_per_step"),
            "kernel_ms_per_step": d["roofline"]["kernel_ms_per_step"], "nn_fallback_fraction": d["roofline"]["nn_fallback_fraction"],
            "roofline_mme": {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms")} if d["roofline"]["kernel"] == "mme" else None,
            "roofline_nn_grid": d["roofline"].get("nn_grid"), "results": d["results"]}


for w in ("campus", "c3_20m", "c5_tunnel"):
    d, _ = bench(["--workload", w])
    out[w] = entry(d)
# density axis: the same 50 M + 50 M multisession pair at 2 500 (bench default), 10^4 (the reference's downsample_size 0.01) and 4 x 10^4 pts/m^2
stats_lib = os.path.join(os.getcwd(), "scratch", "libmapeval_hip_stats.so")
dens = {}
for rho in (2500, 10000, 40000):
    d, _ = bench(["--workload", "c4_multisession", "--density", str(rho), "--no-h2d"])
    e = entry(d)
    if os.path.exists(stats_lib):
        _, err = bench(["--workload", "c4_multisession", "--density", str(rho), "--no-h2d", "--steps", "1", "--warmup", "0", "--no-roofline"],
                       env={"MAPEVAL_HIP_LIB": stats_lib})
        st = []
        for m in re.finditer(r"\[mme stats\] queries=(\d+) pass rounds=(\d+) .*?candidates/round=([\d.]+) queries served/round=([\d.]+) accepted/query=([\d.]+)", err):
            q, rounds, cand, served, acc = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))
            st.append({"queries": q, "rounds_per_wave": rounds / (q / 64.0), "candidates_per_round": cand, "lanes_served_per_round": served,
                       "accepted_per_query": acc, "lane_efficiency": acc * served / (cand * 64.0) if cand else None})
        e["mme_group_stats"] = st[:2]  # est pass, gt pass
        if st:
            pairs = sum(s["accepted_per_query"] * s["queries"] for s in st[:2])
            e["mme_ns_per_accepted_pair"] = e["kernel_ms_per_step"]["mme"] * 1e6 / pairs
    dens[str(rho)] = e
out["density_sweep_c4_multisession"] = dens
out["_note"] = ("python bench.py --workload <w> --cpu-baseline off --steps 3 (profiles/run_workloads.sh); density sweep: --density <rho> on the "
                "c4_multisession pair, group statistics from the -DME_MME_STATS build")
json.dump(out, open(sys.argv[1] + "/workloads.json", "w"), indent=1)
PY
tail -c 600 "$OUT/bench_default.json"
