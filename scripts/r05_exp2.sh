#!/bin/bash
# Round 5 experiments: packed with 2 blocks per launch after the compact-chain change; multi-trait sparse steady state with the quiet XCD
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_exp2; mkdir -p $OUT
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --via-api 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag it/s=%.2f ms=%.3f sweep=%.3f launch_us=%.2f bs=%d m=%d' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['config']['block_size'], d['config']['blocks_per_launch']))" | tee -a $OUT/exp.log; }
run packed --storage packed2bit
JWAS_BENCH_FORCE_PACKED_GROUPS=2 run packed_g2 --storage packed2bit
JWAS_BENCH_FORCE_PACKED_GROUPS=4 run packed_g4 --storage packed2bit
run c4sparse --workload config4 --mt-prior sparse
JWAS_HIP_QUIET_XCD=1 run c4sparse_quiet1 --workload config4 --mt-prior sparse
JWAS_HIP_QUIET_XCD=0 run c4sparse_quiet0 --workload config4 --mt-prior sparse
JWAS_HIP_QUIET_XCD=0 run config3_quiet0 --workload config3
run config3 --workload config3
JWAS_HIP_QUIET_XCD=1 run config2_quiet1
