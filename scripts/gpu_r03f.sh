#!/bin/bash
mkdir -p gpurun_out/r03f
timeout 1500 python -m pytest tests/test_gpu_f64.py -q -x 2>&1 | tail -30 > gpurun_out/r03f/f64_tests.log
tail -30 gpurun_out/r03f/f64_tests.log
