#!/bin/bash
# scratch: the config-4 bench lines again, now that profiles/traffic.json holds their measured traffic
mkdir -p gpurun_out/r03b
python bench.py --via-api 0 --workload config4 --warmup 10 --burnin 0 > gpurun_out/r03b/bench_config4.json 2>/dev/null
python bench.py --via-api 0 --workload config4 --mt-prior sparse > gpurun_out/r03b/bench_config4_sparse.json 2>/dev/null
python bench.py --via-api 0 --workload config4 --mt-method BayesB --warmup 10 --burnin 0 > gpurun_out/r03b/bench_config4_bayesb.json 2>/dev/null
tail -c 700 gpurun_out/r03b/bench_config4.json
