#!/bin/bash
# Round 5: update-role geometry under grouped launches (4 x 1024 markers per launch, 122 us): more streaming workgroups than 224?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_r; mkdir -p $OUT
run() {
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 --groups $1 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groups=$1 budget=${JWAS_HIP_WG_BUDGET:-224} spg=${JWAS_HIP_SPG:-auto} ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" | tee -a $OUT/bench.log
}
run 4
JWAS_HIP_WG_BUDGET=252 JWAS_HIP_SPG=8 run 4
JWAS_HIP_WG_BUDGET=252 JWAS_HIP_SPG=7 run 4
JWAS_HIP_WG_BUDGET=240 JWAS_HIP_SPG=8 run 4
JWAS_HIP_SPG=8 run 4
run 4
JWAS_HIP_WG_BUDGET=252 JWAS_HIP_SPG=8 run 0
JWAS_HIP_WG_BUDGET=252 JWAS_HIP_SPG=7 run 0
