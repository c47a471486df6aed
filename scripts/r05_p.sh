#!/bin/bash
# Round 5: the multi-trait block policy (256 while the chain is dense, 512 once it is sparse) along config 4's chain; e2e MT tests; the Rule T golden
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_e2e.py -q -x -k "golden or three_trait or multi" 2>&1 | tail -4
JWAS_BENCH_VERBOSE=1 JWAS_BENCH_LOG_STATES=100 timeout 1200 python bench.py --no-cpu-baseline --via-api 0 --steps 100 --workload config4 --warmup 0 --burnin 2900 > $OUT/bench_config4_longrun.json 2> $OUT/chain.err
grep "joint-state" $OUT/chain.err | awk 'NR%2==0' | cut -c18-200 > $OUT/config4_chain.log; tail -8 $OUT/config4_chain.log
tail -1 $OUT/bench_config4_longrun.json | cut -c1-250
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --warmup 10 --burnin 0 | cut -c1-200
