#!/bin/bash
# Round 5: the compact candidate chain from fewer candidates on (JWAS_HIP_COMPACT_OFF = 256 n) -- packed, grouped / plain; phase counters
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_z; mkdir -p $OUT
run() { JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 --steps 10 "$@" 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cmin=${JWAS_HIP_COMPACT_OFF:-0} $* sweep=%.3f launch_us=%.2f' % (d['config']['device_sweep_ms'], d['roofline']['avg_launch_us']))" | tee -a $OUT/bench.log; grep "jwas_hip\] blocks" $OUT/err.log | tail -1 | cut -c1-420 | tee -a $OUT/bench.log; }
for c in 0 256 512 768; do
  if [ $c = 0 ]; then unset JWAS_HIP_COMPACT_OFF; else export JWAS_HIP_COMPACT_OFF=$c; fi
  run --storage packed2bit
  JWAS_BENCH_FORCE_PACKED_GROUPS=4 run --storage packed2bit
  run --groups 4
done
