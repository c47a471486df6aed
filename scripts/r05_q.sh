#!/bin/bash
# Round 5: phase counters of config 3 (BayesR) and fixed pi
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_q; mkdir -p $OUT
JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 --workload config3 --warmup 20 --steps 10 > $OUT/c3.json 2> $OUT/c3.log
grep "jwas_hip\] blocks" $OUT/c3.log | tail -2 | cut -c1-700
tail -1 $OUT/c3.json | cut -c1-200
