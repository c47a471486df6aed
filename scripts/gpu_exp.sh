#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "cooperative or dense_big or all_included" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing" | tail -3
for w in "refbench" "refbench --block-size 128" "config4"; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 10 --burnin 0 --no-cpu-baseline --via-api 0 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'it/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'sweep_ms', round(d['config']['device_sweep_ms'],2), 'bs', d['config']['block_size'], 'frac', round(d['roofline']['frac'],3))
"
done
