#!/bin/bash
# Round 6: L2 prefetch of the next block's front in the steady-state grouped kernel: grouped / packed / literal tests + same-box A/B
# against the library before it (csrc/_dev/libjwas_hip_front2.so).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -3
B="--no-cpu-baseline --via-api 0"
run() {  # label, env, args
  env $2 python bench.py $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(12), '$3'.ljust(56), 'it/s=%.2f ms=%.3f dev_ms=%.3f launch_us=%.2f ev=%.0f' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['config']['events_per_sweep']))" 2>&1 | tail -1
}
for v in new front2 new front2; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  run $v X=1 "--storage packed2bit"
  run $v X=1 "--workload config2"
done
cp /tmp/new.so $L
run new-off JWAS_HIP_COMPACT_OFF=16 "--storage packed2bit"
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --storage packed2bit --steps 3 2>&1 | grep "jwas_hip\] blocks" | tail -1 | cut -c1-330
