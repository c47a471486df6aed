#!/bin/bash
# Round 6: (1) the one test that failed under JWAS_HIP_PINGPONG=1 in r06_final3, alone, three times; (2) the multi-trait skip-and-verify
# pass (sampler_role_mt): its parity test, every multi-trait GPU test, the fuzz; (3) same-box A/B of the working tree (plain group correction
# in the steady-state kernel + skip and verify) against HEAD (csrc/_dev/libjwas_hip_head.so) and the round-5 library.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
for i in 1 2 3; do JWAS_HIP_PINGPONG=1 timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "row_shards_single_rank" 2>&1 | grep -v "$F" | tail -15; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "skip_and_verify" 2>&1 | grep -v "$F" | tail -25
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py tests/test_gpu_packed.py tests/test_gpu_statistical.py tests/test_gpu_f64.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -8
B="--no-cpu-baseline --via-api 0"
run() {  # label, env, args
  env $2 python bench.py $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(12), '$3'.ljust(52), 'it/s=%.2f ms=%.3f dev_ms=%.3f launch_us=%.2f ev=%.0f' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['config']['events_per_sweep']))" 2>&1 | tail -1
}
for v in new head new head r05; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  run $v X=1 "--workload config4 --mt-prior sparse"
  run $v X=1 "--storage packed2bit"
  run $v X=1 "--workload config2"
done
cp /tmp/new.so $L
run new-noskip JWAS_HIP_COMPACT_OFF=8 "--workload config4 --mt-prior sparse"
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --workload config4 --mt-prior sparse --steps 5 2>&1 | grep "jwas_hip\] blocks" | tail -2 | cut -c1-330
JWAS_HIP_COMPACT_OFF=8 JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --workload config4 --mt-prior sparse --steps 5 2>&1 | grep "jwas_hip\] blocks" | tail -2 | cut -c1-330
