#!/bin/bash
# Round 5 experiment: temporal (cache-allocating) column loads in the update role (-DJWAS_EXP_TEMPORAL, csrc/_dev) against the
# non-temporal ones of the shipped build: do dense sweeps find the re-read columns in the Infinity Cache?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_o; mkdir -p $OUT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/ship.so
for v in ship temporal ship temporal; do
  if [ $v = temporal ]; then cp jwas.jl_amd/csrc/_dev/libjwas_hip.so $L; else cp /tmp/ship.so $L; fi
  for w in refbench config4 config2; do
    X="--warmup 10 --burnin 0"; [ $w = config2 ] && X=""
    python bench.py --no-cpu-baseline --via-api 0 --workload $w $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w ms=%.2f sweep=%.2f launch_us=%.2f' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us']))"
  done
done
cp /tmp/ship.so $L
