#!/bin/bash
# round 3, first GPU call: the whole GPU suite on the new host plumbing + the default bench line
mkdir -p gpurun_out/r03a
export JWAS_BENCH_VERBOSE=0
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03a/gpu_tests.log
tail -8 gpurun_out/r03a/gpu_tests.log
