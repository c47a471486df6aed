#!/bin/bash
# Round 5: k_group_step with per-block sampler arguments from the host (fewer spills): parity, then dense / packed timings + phase counters
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_exp3; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_groups.py -x -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -3 | tee $OUT/tests.log
run() { tag=$1; shift; JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 "$@" 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag it/s=%.2f ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f m=%d' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['blocks_per_launch']))" | tee -a $OUT/exp.log; grep "jwas_hip\] blocks" $OUT/err.log | tail -1 | grep -o "role=[0-9]*" | tee -a $OUT/exp.log; }
run default
run groups0 --groups 0
run default
JWAS_BENCH_FORCE_PACKED_GROUPS=4 run packed_g4 --storage packed2bit
JWAS_BENCH_FORCE_PACKED_GROUPS=2 run packed_g2 --storage packed2bit
run packed --storage packed2bit
