#!/bin/bash
# Round 6 experiment: the cooperative apply (update_role COOP) for the merged list of grouped launches.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_q; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
JWAS_HIP_GROUP_COOP=1 timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -3
B="--no-cpu-baseline --via-api 0 --steps 20"
for v in 0 1 0 1; do
  for w in "--workload config3" "--workload config2 --pi-fixed 0.95"; do
    JWAS_HIP_GROUP_COOP=$v python bench.py $B $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop=$v', '$w'.ljust(36), 'it/s=%.2f ms=%.2f launch_us=%.2f m=%d' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['blocks_per_launch']))" 2>&1 | tail -1
  done
done
