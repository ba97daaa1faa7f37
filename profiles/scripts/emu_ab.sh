#!/bin/bash
for i in 1 2; do
echo "classic protocol (ME_DIST_LEAN=0)"; ME_DIST_LEAN=0 python profiles/scripts/emu_classic.py 50000000 --workload c4_multisession --worlds 8 2>/dev/null | tail -1 | python -c "
import json,sys; s=json.load(sys.stdin); w=s['per_world']['8']; print('  ', w['per_rank_ms']); print('  ', {a:b for a,b in w['slowest_rank_kernel_ms'].items() if isinstance(b,float)})"
echo "lean"; python profiles/emulate_scaling.py 50000000 --workload c4_multisession --worlds 8 2>/dev/null | tail -1 | python -c "
import json,sys; s=json.load(sys.stdin); w=s['per_world']['8']; print('  ', w['per_rank_ms']); print('  ', {a:b for a,b in w['slowest_rank_kernel_ms'].items() if isinstance(b,float)})"
done
echo "trace lean rank 0"; ME_DIST_TRACE=1 python profiles/emulate_scaling.py 50000000 --workload c4_multisession --worlds 8 2>&1 | grep -i "trace\|rank 0" | tail -4
echo "trace classic rank 0"; ME_DIST_LEAN=0 ME_DIST_TRACE=1 python profiles/scripts/emu_classic.py 50000000 --workload c4_multisession --worlds 8 2>&1 | grep -i "trace\|rank 0" | tail -4
