#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tl; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/bench.py --cpu-baseline off --no-h2d --steps 3 --warmup 1 --settle-max 2 --no-roofline > /dev/null 2> $OUT/err.txt
cd $ROOT
python - $OUT <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]); n = re.sub(r"<.*", "", n).split("::")[-1]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n))
rows.sort()
mort = [i for i, r in enumerate(rows) if r[3] == "k_morton"]
a, b = mort[-4], mort[-2]
step = rows[a:b]; t0 = step[0][0]; t1 = max(r[1] for r in step)
print("two-lane step: %.2f ms wall, %d kernels" % ((t1 - t0) / 1e6, len(step)))
qs = sorted({r[2] for r in step})
for q in qs:
    ks = [r for r in step if r[2] == q]
    print("queue %s: %d kernels, busy %.2f ms" % (q, len(ks), sum(r[1]-r[0] for r in ks)/1e6))
# every kernel of the main queue (the one with k_mme3) with the gap before it
mainq = [r for r in step if r[3] == "k_mme3"][0][2]
prev = None
for r in step:
    if r[2] != mainq: continue
    gap = (r[0] - prev) / 1e3 if prev else 0
    if r[1]-r[0] > 80_000 or gap > 100: print("  %7.3f  dur %8.3f  gap %7.1f us  %s" % ((r[0]-t0)/1e6, (r[1]-r[0])/1e6, gap, r[3]))
    prev = r[1]
for q in qs:
    if q == mainq: continue
    print("--- queue %s, kernels > 0.08 ms" % q)
    for r in step:
        if r[2] != q: continue
        if r[1]-r[0] > 80_000:
            print("  %7.3f  dur %8.3f  %s" % ((r[0]-t0)/1e6, (r[1]-r[0])/1e6, r[3]))
PY
find $OUT -name "*.csv" -delete
