#!/bin/bash
# Round 5: grouped launches on the 512-marker blocks of high-turnover sweeps (config 3 BayesR, fixed pi)?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_s; mkdir -p $OUT
run() {
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 "$@" 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* small=${JWAS_BENCH_GROUPS_SMALL:-0} ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f bs=%d m=%d' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['block_size'], d['config']['blocks_per_launch']))" | tee -a $OUT/bench.log
}
run --workload config3
JWAS_BENCH_GROUPS_SMALL=2 run --workload config3
JWAS_BENCH_GROUPS_SMALL=4 run --workload config3
run --pi-fixed 0.95
JWAS_BENCH_GROUPS_SMALL=2 run --pi-fixed 0.95
JWAS_BENCH_GROUPS_SMALL=4 run --pi-fixed 0.95
