#!/bin/bash
# A/B of the 1-NN grid kernels (round 3): VALU ranking (V=1) vs MFMA ranking (V=2) at 8 / 5 waves per SIMD; -DME_AB build
export MAPEVAL_HIP_LIB=$PWD/scratch/libmapeval_hip_ab.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -x -q -m gpu -k "nn1 or stats or chamfer or c1_process or 1m_three or c2_5m" 2>&1 | tail -4
run() { python bench.py --cpu-baseline off --no-h2d --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step', round(d['ms_per_step'],2), {k:round(v,2) for k,v in r['kernel_ms_per_step'].items() if k in ('nn_grid','nn1','mme')}, 'fallback', r['nn_fallback_fraction'], 'CD', d['results']['CD'], 'AC', d['results']['AC'][0])"; }
for cfg in "1 8" "2 8" "2 5"; do set -- $cfg; echo "== nn_grid V=$1 W=$2"; ME_NN_GRID_V=$1 ME_NN_GRID_WAVES=$2 run; done
