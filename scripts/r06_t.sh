#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_t; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
JWAS_HIP_PINGPONG=1 timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -2
timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_e2e.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -2
B="--no-cpu-baseline --via-api 0 --steps 20"
for coop in 0 1; do
for w in "--workload config3" "--workload config2 --pi-fixed 0.95"; do
  JWAS_HIP_GROUP_COOP=$coop JWAS_HIP_DEBUG_PHASES=1 python bench.py $B $w 2> $OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop=$coop', '$w'.ljust(36), 'it/s=%.2f ms=%.2f launch_us=%.2f m=%d frac=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['blocks_per_launch'], d['roofline']['frac']))" 2>&1 | tail -1
  grep "jwas_hip\] blocks" $OUT/err.log | tail -1 | grep -o "compact: blocks=[0-9]* fallback=[0-9]*\|last_workgroup=[0-9]*"
done; done
