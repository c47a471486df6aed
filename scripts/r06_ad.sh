#!/bin/bash
# Round 6: explicit wait in front of the multi-trait front (all of its ~60 loads one flight): multi-trait tests + same-box A/B (csrc/_dev/libjwas_hip_front2.so).
# partial sums straight-line): multi-trait tests + same-box A/B against the library before it (csrc/_dev/libjwas_hip_front2.so).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py tests/test_gpu_statistical.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -6
B="--no-cpu-baseline --via-api 0"
run() {  # label, env, args
  env $2 python bench.py $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(12), '$3'.ljust(52), 'it/s=%.2f ms=%.3f dev_ms=%.3f launch_us=%.2f ev=%.0f' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['config']['events_per_sweep']))" 2>&1 | tail -1
}
for v in new front2 new front2; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  run $v X=1 "--workload config4 --mt-prior sparse"
  run $v X=1 "--workload config4"
  run $v X=1 "--workload config4 --mt-method BayesB"
done
cp /tmp/new.so $L
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --workload config4 --mt-prior sparse --steps 5 2>&1 | grep "jwas_hip\] blocks" | tail -1 | cut -c1-330
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --workload config4 --steps 5 2>&1 | grep "jwas_hip\] blocks" | tail -1 | cut -c1-330
