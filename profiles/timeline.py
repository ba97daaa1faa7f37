"""Gap analysis of one suite step from a rocprofv3 kernel trace (run on the GPU box):
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline
   python profiles/timeline.py gpurun_out/tl
Prints, for the last step, the busy / idle time of the queue that runs k_mme and the kernels longer than 0.2 ms."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
mme = [r for r in rows if "k_mme" in r[3] and "final" not in r[3] and "unpermute" not in r[3]]
q_main = mme[-1][2]
# last step = from the last-but-one k_mme of the EST cloud backwards to the first kernel after the previous step's end
last_two = mme[-2:]
t_end = max(r[1] for r in rows)
# start of the last step: the first kernel after the end of the third-last k_mme's step; approximate with the big idle
# gap (synthetic data is resident, so a step starts with the slab-free upload kernels)
t0 = last_two[0][0]
prev = [r for r in rows if r[1] < t0]
# walk back until the previous step's final kernels (a gap > 0.3 ms on every queue marks the barrier between steps)
start = t0
for r in reversed(prev):
    if start - r[1] > 300_000:
        break
    start = min(start, r[0])
step = [r for r in rows if r[0] >= start]
print("step window %.2f ms, main queue %s" % ((t_end - start) / 1e6, q_main))
for q in sorted({r[2] for r in step}):
    ks = sorted(r for r in step if r[2] == q)
    busy = sum(r[1] - r[0] for r in ks)
    print("queue %s: %d kernels, busy %.2f ms, first %.2f last %.2f" % (q, len(ks), busy / 1e6, (ks[0][0] - start) / 1e6, (ks[-1][1] - start) / 1e6))
ks = sorted(r for r in step if r[2] == q_main)
cur = start
gaps = []
for r in ks:
    if r[0] - cur > 50_000:
        gaps.append(((cur - start) / 1e6, (r[0] - cur) / 1e6, r[3][:50]))
    cur = max(cur, r[1])
print("gaps > 0.05 ms on the main queue (at ms, length ms, next kernel):")
for g in gaps:
    print("  %.2f  %.3f  %s" % g)
print("kernels > 0.2 ms:")
for r in sorted(step):
    if r[1] - r[0] > 200_000:
        print("  %.2f  %.2f ms  q%s  %s" % ((r[0] - start) / 1e6, (r[1] - r[0]) / 1e6, r[2], r[3][:60]))
