#!/bin/bash
L=cloud_map_evaluation_amd/libmapeval_hip.so
cp $L /tmp/tree.so
one() { python profiles/host_one_call.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', [r['metric_phase_ms'] for r in d['runs']]); [print('      ', r['stages']) for r in d['runs']]"; }
for i in 1 2; do
  echo "A ${A:-prev}"; cp ${A:-scratch/libmapeval_hip_prev.so} $L; one
  echo "B tree"; cp /tmp/tree.so $L; one
done
