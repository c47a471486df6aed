#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_k; mkdir -p $OUT
B="--no-cpu-baseline --via-api 0"
JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B --storage packed2bit --steps 10 > $OUT/bench_packed.json 2> $OUT/bench_packed.log
grep "jwas_hip\] blocks" $OUT/bench_packed.log | tail -1 | cut -c1-300; grep "jwas_hip\] blocks" $OUT/bench_packed.log | tail -1 | grep -o "group:.*"
JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B --steps 10 > $OUT/bench_config2.json 2> $OUT/bench_config2.log
grep "jwas_hip\] blocks" $OUT/bench_config2.log | tail -1 | grep -o "role=[0-9]*\|group:.*"
