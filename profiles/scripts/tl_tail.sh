#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tlt; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/bench.py --cpu-baseline off --no-h2d --steps 3 --warmup 1 --settle-max 2 --no-roofline > /dev/null 2> $OUT/err.txt
cd $ROOT
python - $OUT <<'PY' > $OUT/tail.txt
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]); n = re.sub(r"<.*", "", n).split("::")[-1]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n, r.get("Grid_Size","?"), r.get("Workgroup_Size","?")))
rows.sort()
mort = [i for i, r in enumerate(rows) if r[3] == "k_morton"]
for (a, b) in ((mort[-4], mort[-2]), (mort[-6], mort[-4])):
    step = rows[a:b]; t0 = step[0][0]; t1 = max(r[1] for r in step)
    print("two-lane step: %.2f ms wall, %d kernels" % ((t1 - t0) / 1e6, len(step)))
    for r in step:
        if (r[0]-t0)/1e6 < 40.5: continue
        print("  q%s %7.3f  dur %8.3f  end %7.3f  %s grid=%s wg=%s" % (r[2], (r[0]-t0)/1e6, (r[1]-r[0])/1e6, (r[1]-t0)/1e6, r[3], r[4], r[5]))
PY
find $OUT -name "*.csv" -delete
