"""Timeline of one TWO-LANE step (the last timed one, before the per-kernel timing pass) from a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --cpu-baseline off --no-h2d --steps 3 --warmup 1
   python profiles/timeline2.py $OUT
Per queue: busy time, first / last kernel; the union of both queues (time the GPU ran nothing); the long kernels in order."""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = re.sub(r"\(.*", "", r["Kernel_Name"])
            n = re.sub(r"<.*", "", n).split("::")[-1]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n))
rows.sort()
mort = [i for i, r in enumerate(rows) if r[3] == "k_morton"]
# steps: every step starts with two k_morton (one per cloud); the last pair belongs to the timing pass
a, b = mort[-4], mort[-2]
step = rows[a:b]
t0 = step[0][0]
t1 = max(r[1] for r in step)
print("two-lane step: %.2f ms wall, %d kernels" % ((t1 - t0) / 1e6, len(step)))
for q in sorted({r[2] for r in step}):
    ks = [r for r in step if r[2] == q]
    print("queue %s: %d kernels, busy %.2f ms, first at %.2f, last ends %.2f" % (q, len(ks), sum(r[1] - r[0] for r in ks) / 1e6, (ks[0][0] - t0) / 1e6, (max(r[1] for r in ks) - t0) / 1e6))
ev = sorted((r[0], r[1]) for r in step)
cur, idle, gaps = t0, 0, []
for s, e in ev:
    if s > cur:
        idle += s - cur
        if s - cur > 30_000:
            gaps.append(((cur - t0) / 1e6, (s - cur) / 1e6))
    cur = max(cur, e)
print("GPU idle (no kernel on any queue): %.2f ms; gaps > 0.03 ms: %s" % (idle / 1e6, ", ".join("%.2f@%.1f" % (g[1], g[0]) for g in gaps)))
print("kernels > 0.3 ms (start ms, duration ms, queue, name):")
for r in step:
    if r[1] - r[0] > 300_000:
        print("  %6.2f  %6.2f  q%s  %s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[2], r[3]))
