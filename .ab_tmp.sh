python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_host.py -q -x 2>&1 | tail -12
python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernel_ms_per_step'].items()}, d['results'])"
