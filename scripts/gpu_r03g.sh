#!/bin/bash
mkdir -p gpurun_out/r03g
timeout 2700 python -m pytest tests -m gpu -q -x -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 > gpurun_out/r03g/gpu_tests.log
tail -25 gpurun_out/r03g/gpu_tests.log
for w in "refbench" "refbench --block-size 256" "refbench --block-size 128" "config2 --pi-fixed 0.0"; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --workload $w --steps 10 --warmup 10 --burnin 0 --no-cpu-baseline --via-api 0 2> gpurun_out/r03g/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'it/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'sweep_ms', round(d['config']['device_sweep_ms'],2), 'bs', d['config']['block_size'], 'frac', round(d['roofline']['frac'],3))
"
tail -1 gpurun_out/r03g/err.txt
done
