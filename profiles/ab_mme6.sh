#!/bin/bash
# A/B of the MME kernels (round 3): parity tests on the default library, then timings of the variants of the -DME_AB build
python -m pytest tests/test_gpu_parity.py tests/test_gpu_degenerate.py tests/test_gpu_ref.py -x -q -m gpu -k "mme or degenerate or tunnel or c1_process or 1m_three" 2>&1 | tail -5
run() { python bench.py --cpu-baseline off --no-h2d --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step', round(d['ms_per_step'],2), 'kernel ms', {k:round(v,2) for k,v in r['kernel_ms_per_step'].items()}, 'MMEvalid', d['results']['MME_valid'], 'MME', d['results']['MME_est'], d['results']['MME_gt'])"; }
export MAPEVAL_HIP_LIB=$PWD/scratch/libmapeval_hip_ab.so
for cfg in "3 8 32" "6 8 32" "6 6 32" "6 8 64"; do set -- $cfg; echo "== V=$1 W=$2 T=$3"; ME_MME_V=$1 ME_MME_WAVES=$2 ME_MME_TILE=$3 run; done
export MAPEVAL_HIP_LIB=$PWD/scratch/libmapeval_hip_stats.so
echo "== stats V=6"; ME_MME_V=6 python bench.py --cpu-baseline off --no-h2d --no-roofline --steps 1 --warmup 0 2>&1 | grep "mme stats" | head -4
echo "== stats V=3"; ME_MME_V=3 python bench.py --cpu-baseline off --no-h2d --no-roofline --steps 1 --warmup 0 2>&1 | grep "mme stats" | head -4
