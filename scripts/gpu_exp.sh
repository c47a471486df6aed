#!/bin/bash
mkdir -p gpurun_out/r03f64
timeout 600 python -m pytest tests/test_gpu_f64.py -q -x 2>&1 | tail -2
timeout 1200 python scripts/f64_bench.py 20000 20000 2>/dev/null | tail -1 | tee gpurun_out/r03f64/f64_bench.json
