#!/bin/bash
run() { MAPEVAL_HIP_LIB=$1 python bench.py --cpu-baseline off --no-h2d --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('  ms/step %.2f' % d['ms_per_step'], '| mme %.2f nn_grid %.2f' % (k['mme'], k['nn_grid']), '| MME', d['results']['MME_est'], d['results']['MME_valid'])"; }
for w in 3 4 6; do echo "waves $w"; run $PWD/scratch/libmapeval_hip_w$w.so; done
echo "waves 8 (tree)"; run $PWD/cloud_map_evaluation_amd/libmapeval_hip.so
python profiles/host_one_call.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('host_one_call', [r['metric_phase_ms'] for r in d['runs']], [r['stages'] for r in d['runs']])"
