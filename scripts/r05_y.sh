#!/bin/bash
# Round 5: role split of the packed sweep and of the grouped headline (development build: JWAS_HIP_DEBUG_ROLE=1 update only, 2 sampler only; results wrong)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_y; mkdir -p $OUT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/ship.so; cp jwas.jl_amd/csrc/_dev/libjwas_hip.so $L
run() { timeout 600 python bench.py --no-cpu-baseline --via-api 0 --steps 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('role=${JWAS_HIP_DEBUG_ROLE:-both} $* sweep=%.3f launch_us=%.2f' % (d['config']['device_sweep_ms'], d['roofline']['avg_launch_us']))" | tee -a $OUT/roles.log; }
for r in 0 1 2; do
  export JWAS_HIP_DEBUG_ROLE=$r
  run --storage packed2bit
  run --groups 0
done
unset JWAS_HIP_DEBUG_ROLE
cp /tmp/ship.so $L
