#!/bin/bash
# SQ counters per wavefront of the MME kernel in the product library (round 4: k_mme7): bash profiles/sq_mme7.sh [tag]
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --cpu-baseline off --no-h2d --workload campus --points 10000000 --steps 1 --warmup 0 --no-roofline"
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_sq_a" -- $BENCH > /dev/null 2> "$OUT/a.err"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq_b" -- $BENCH > /dev/null 2> "$OUT/b.err"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/pmc_sq_c" -- $BENCH > /dev/null 2> "$OUT/c.err"
cd "$ROOT"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/pmc_sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0]
        if "mme" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {}
for k in acc:
    waves = acc[k].get("SQ_WAVES", 0) / max(1, cnt[k].get("SQ_WAVES", 1))
    res[k] = {c: (acc[k][c] / cnt[k][c]) for c in acc[k]}
    res[k]["launches"] = cnt[k].get("SQ_WAVES", 0)
    if waves:
        res[k]["per_wave"] = {c: res[k][c] / waves for c in res[k] if c.startswith("SQ_") and c != "SQ_WAVES" and c != "SQ_BUSY_CYCLES"}
json.dump(res, open(out + "/../" + tag + "_mme_sq_per_wave.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True)[:4000])
PY
for e in "$OUT"/*.err; do echo "== $e"; grep -v "simple_timer\|Opened result file" "$e" | tail -2 | cut -c1-200; done
rm -rf "$OUT"
