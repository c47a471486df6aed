#!/bin/bash
# Round 5 experiment: register batches in flight per wave (1 = shipped, 2, 4) in the update role of grouped launches
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_depth; mkdir -p $OUT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/ship.so
run() { timeout 600 python bench.py --no-cpu-baseline --via-api 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 it/s=%.2f ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" | tee -a $OUT/depth.log; }
for v in ship d2 d4 ship d2 d4; do
  if [ $v = ship ]; then cp /tmp/ship.so $L; else cp jwas.jl_amd/csrc/_exp/libjwas_hip_$v.so $L; fi
  run $v
done
cp /tmp/ship.so $L
