#!/bin/bash
# Round 5, final code: long differential fuzz (whole sweep + Rule T + grouped launches), device vs oracle
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_fuzz_final; mkdir -p $OUT
( time JWAS_FUZZ_CASES=12000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_groups.py -q -n 8 -k "random" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) > $OUT/fuzz_final.log 2>&1
tail -6 $OUT/fuzz_final.log
