#!/bin/bash
# the long differential fuzz run: JWAS_FUZZ_CASES random configurations, device vs oracle BIT FOR BIT
mkdir -p gpurun_out/r03fuzz
export JWAS_FUZZ_CASES=${1:-3000}
( time timeout 3500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 -k random_configurations 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 ) > gpurun_out/r03fuzz/fuzz_${JWAS_FUZZ_CASES}.log 2>&1
tail -8 gpurun_out/r03fuzz/fuzz_${JWAS_FUZZ_CASES}.log
