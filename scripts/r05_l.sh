#!/bin/bash
# Round 5: where a config-4 chain (reference default prior, pi estimated) goes: joint-state counts and sweep time every 100 sweeps
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_l; mkdir -p $OUT
JWAS_BENCH_VERBOSE=1 JWAS_BENCH_LOG_STATES=100 timeout 1200 python bench.py --no-cpu-baseline --via-api 0 --steps 100 --workload config4 --warmup 0 --burnin ${BURN:-1700} > $OUT/chain.json 2> $OUT/chain.log
grep "joint-state\|joint state" $OUT/chain.log | cut -c1-220
tail -1 $OUT/chain.json | cut -c1-300
