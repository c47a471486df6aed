#!/bin/bash
mkdir -p gpurun_out/r03e
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r03e/gpu_tests.log
tail -6 gpurun_out/r03e/gpu_tests.log
for w in "config4 --mt-method BayesB" "config4 --mt-method BayesB --mt-prior sparse"; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 10 --burnin 0 --no-cpu-baseline --via-api 0 2> gpurun_out/r03e/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'it/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'sweep_ms', round(d['config']['device_sweep_ms'],2), 'bs', d['config']['block_size'], 'frac', round(d['roofline']['frac'],3))
"
tail -3 gpurun_out/r03e/err.txt
done
