#!/bin/bash
run() { MAPEVAL_HIP_LIB=$1 python bench.py --cpu-baseline off --no-h2d --steps 10 --warmup 3 --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  ms/step %.2f' % d['ms_per_step'], d.get('step_ms'))"; }
for i in 1 2; do for l in "$@"; do echo $l; run $PWD/$l; done; done
