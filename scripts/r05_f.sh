#!/bin/bash
# the whole GPU suite on the current library
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_f; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -40 > $OUT/gpu_tests.log
tail -40 $OUT/gpu_tests.log | cut -c1-250
