#!/bin/bash
# Round 6: helper waves load their sub-block's state before they wait for their turn: tests + same-box A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "skip_and_verify or three_resident or mt_" 2>&1 | grep -v "$F" | tail -3
JWAS_FUZZ_CASES=1200 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -n 8 -k "skip_and_verify" 2>&1 | grep -v "$F" | tail -3
B="--no-cpu-baseline --via-api 0"
run() {  # label, env, args
  env $2 python bench.py $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(12), '$3'.ljust(48), 'it/s=%.2f ms=%.3f dev_ms=%.3f launch_us=%.2f bs=%d' % (d['value'], d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['config']['block_size']))" 2>&1 | tail -1
}
for v in new front2 new front2 new front2; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  run $v X=1 "--workload config4 --mt-prior sparse"
done
cp /tmp/new.so $L
