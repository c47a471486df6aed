#!/bin/bash
# Round 5, final code, ONE box: A/B of the round's two schedule changes (grouped launches; compact chain from one candidate on)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_ab; mkdir -p $OUT
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --via-api 0 "$@" 2>/dev/null > $OUT/bench_$tag.json; python -c "
import json
d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag it/s=%.2f ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" | tee -a $OUT/ab.log; }
for rep in 1 2; do
  run default_$rep
  run groups0_$rep --groups 0
  run packed_$rep --storage packed2bit
  JWAS_HIP_COMPACT_OFF=1536 run packed_cmin6_$rep --storage packed2bit
done
