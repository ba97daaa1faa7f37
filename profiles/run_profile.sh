#!/bin/bash
# Collects the rocprofv3 evidence behind profiles/<tag>_*: run on the GPU box from the repo root,
#   bash profiles/run_profile.sh r01
# Counters are collected in their own passes (never together with --kernel-trace/--stats).
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --cpu-baseline off --no-h2d"
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 3 --warmup 1 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
# the same with ONE lane (round 5): per-launch averages that are not stretched by the other lane's kernels — what
# roofline.avg_launch_ms (HIP events, single-lane pass of bench.py) can be recomputed from
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -- $BENCH --steps 3 --warmup 1 --no-overlap --no-roofline > "$OUT/bench_single_lane_under_rocprof.json" 2> "$OUT/stats1.err"
timeout 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- $BENCH --settle-max 0 --steps 1 --warmup 0 --no-roofline > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -- $BENCH --settle-max 0 --steps 1 --warmup 0 --no-roofline > /dev/null 2> "$OUT/pmc_write.err"
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_sq_a" -- $BENCH --settle-max 0 --workload campus --points 10000000 --steps 1 --warmup 0 --no-roofline > /dev/null 2> "$OUT/pmc_sq_a.err"
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq_b" -- $BENCH --settle-max 0 --workload campus --points 10000000 --steps 1 --warmup 0 --no-roofline > /dev/null 2> "$OUT/pmc_sq_b.err"
cd "$ROOT"
# keep only what the summaries need (raw traces are large)
find "$OUT" -name "*_kernel_trace.csv" -delete 2>/dev/null
python profiles/summarize.py "$OUT" "$OUT/summary" "$TAG" | tail -40
python profiles/summarize.py "$OUT" "$OUT/summary_single_lane" "$TAG" stats1 | tail -12
# gpurun copies back at most 64 MiB: keep the summaries, drop the raw collections
mkdir -p "$ROOT/gpurun_out/summary_$TAG"
cp "$OUT"/summary_* "$OUT/bench_under_rocprof.json" "$OUT/bench_single_lane_under_rocprof.json" "$ROOT/gpurun_out/summary_$TAG/" 2>/dev/null
for e in "$OUT"/*.err; do echo "== $e"; grep -v "simple_timer\|Opened result file" "$e" | tail -3 | cut -c1-300; done
rm -rf "$OUT"
