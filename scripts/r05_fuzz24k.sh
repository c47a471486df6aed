#!/bin/bash
# Round 5, final code: the long differential fuzz with 24 000 + 3 000 (Rule T) random configurations, device vs oracle bit for bit
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_fuzz; mkdir -p $OUT
( time JWAS_FUZZ_CASES=24000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -n 8 -k "random" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) > $OUT/fuzz_24000_cases.log 2>&1
tail -6 $OUT/fuzz_24000_cases.log
