#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lk; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- env PYTHONPATH=$ROOT python $ROOT/profiles/emulate_scaling.py 50000000 --workload c4_multisession --worlds 8 > /dev/null 2> $OUT/err.txt
cd $ROOT
python - $OUT <<'PY'
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if re.search(r"lattice|plan_|halo_split|split_|k_bbox|k_vox_records|k_gather", n):
            print("%-60s calls %6s avg %8.1f us total %8.2f ms" % (re.sub(r"\(.*", "", n)[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $OUT -name "*.csv" -delete
