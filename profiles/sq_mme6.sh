#!/bin/bash
# SQ counters per wavefront of k_mme6 (the MFMA kernel of round 3) and k_mme3, 10 M-point campus pair, the -DME_AB -DME_MME_STATS build
export TMPDIR=/tmp MAPEVAL_HIP_LIB=$PWD/scratch/libmapeval_hip_stats.so
ROOT=$PWD; OUT=$ROOT/gpurun_out/sq_mme6; rm -rf $OUT; mkdir -p $OUT
BENCH="python $ROOT/bench.py --cpu-baseline off --no-h2d --workload campus --points 10000000 --steps 1 --warmup 0 --no-roofline"
cd /tmp
for v in 3 6; do
  ME_MME_V=$v timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/v$v -- $BENCH > /dev/null 2> $OUT/v$v.err
  ME_MME_V=$v timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/w$v -- $BENCH > /dev/null 2>> $OUT/v$v.err
  grep "mme stats" $OUT/v$v.err | head -2
done
cd $ROOT
python - <<'PY'
import csv, glob, json
from collections import defaultdict
out = {}
for v in (3, 6):
    acc = defaultdict(float)
    for sub in ("v", "w"):
        for f in glob.glob(f"gpurun_out/sq_mme6/{sub}{v}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_mme" in r["Kernel_Name"] and "unpermute" not in r["Kernel_Name"] and "final" not in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"])
    w = acc.get("SQ_WAVES", 1.0) or 1.0
    out[f"k_mme{v}"] = {k: round(x / w, 2) for k, x in acc.items() if k != "SQ_WAVES"}
    out[f"k_mme{v}"]["SQ_WAVES"] = w
json.dump(out, open("gpurun_out/sq_mme6/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
