#!/bin/bash
# Round 6: skip and verify on 1024-marker multi-trait blocks (16 sub-blocks, up to three per helper wave): parity test, multi-trait tests,
# config 4's sparse regime on 512- vs 1024-marker blocks.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "skip_and_verify" 2>&1 | grep -v "$F" | tail -15
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py tests/test_gpu_statistical.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -4
B="--no-cpu-baseline --via-api 0"
run() {  # label, env, args
  env $2 python bench.py $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(12), '$3'.ljust(60), 'it/s=%.2f ms=%.3f dev_ms=%.3f launch_us=%.2f ev=%.0f' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['config']['events_per_sweep']))" 2>&1 | tail -1
}
for i in 1 2; do
  run new X=1 "--workload config4 --mt-prior sparse --block-size 512"
  run new X=1 "--workload config4 --mt-prior sparse --block-size 1024"
  run noskip JWAS_HIP_COMPACT_OFF=8 "--workload config4 --mt-prior sparse --block-size 1024"
done
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --workload config4 --mt-prior sparse --block-size 1024 --steps 5 2>&1 | grep "jwas_hip\] blocks" | tail -1 | cut -c1-330
