#!/bin/bash
# SQ counters per wavefront of the kernels whose name contains PATTERN: bash profiles/sq_kernel.sh PATTERN TAG -- command ...
PAT=$1; TAG=$2; shift 3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="$*"
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_sq_a" -- bash -c "cd $ROOT && $CMD" > /dev/null 2> "$OUT/a.err"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq_b" -- bash -c "cd $ROOT && $CMD" > /dev/null 2> "$OUT/b.err"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM \
    --output-format csv -d "$OUT/pmc_sq_c" -- bash -c "cd $ROOT && $CMD" > /dev/null 2> "$OUT/c.err"
cd "$ROOT"
python - "$OUT" "$TAG" "$PAT" <<'PY'
import csv, glob, json, sys, collections
out, tag, pat = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k in acc:
    # the LONGEST launch of the kernel (by wave cycles / first counter): the est -> gt direction
    res[k] = {c: max(v) for c, v in acc[k].items()}
    res[k]["launches"] = len(acc[k].get("SQ_WAVES", []))
    waves = res[k].get("SQ_WAVES", 0)
    if waves:
        res[k]["per_wave"] = {c: res[k][c] / waves for c in res[k] if c.startswith("SQ_") and c not in ("SQ_WAVES", "SQ_BUSY_CYCLES")}
json.dump(res, open(out + "/../" + tag + "_sq_per_wave.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True)[:6000])
PY
for e in "$OUT"/*.err; do echo "== $e"; grep -v "simple_timer\|Opened result file" "$e" | tail -2 | cut -c1-200; done
rm -rf "$OUT"
