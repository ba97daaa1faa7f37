#!/bin/bash
# A/B of the whole two-lane step: library A (default: scratch/libmapeval_hip_r05.so = the previous round's HEAD) against the tree's
# library, alternating on ONE box:   bash profiles/ab_step.sh [libA.so] [rounds] [extra bench args...]
A=${1:-$PWD/scratch/libmapeval_hip_r05.so}; R=${2:-2}; shift 2 2>/dev/null
B=$PWD/cloud_map_evaluation_amd/libmapeval_hip.so
run() { MAPEVAL_HIP_LIB=$1 python bench.py --cpu-baseline off --no-h2d --steps 10 --warmup 3 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
k=r.get('kernel_ms_per_step',{})
x=d.get('cross_checks',{})
print('  ms/step %.2f' % d['ms_per_step'], 'steps', d.get('step_ms'), '| kernels', {n:round(v,2) for n,v in k.items()}, '| sum %.2f' % sum(k.values()),
      '| x', {n:round(v['ms_per_step'],2) for n,v in x.items() if isinstance(v,dict)}, '| CD', d['results']['CD'], 'MME', d['results']['MME_est'], d['results']['MME_valid'])"; }
for i in $(seq $R); do echo "A $(basename $A)"; run $A "$@"; echo "B tree"; run $B "$@"; done
