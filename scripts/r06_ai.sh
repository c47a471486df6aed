#!/bin/bash
# Round 6, last call: multi-trait test files on the last library + config 4's whole chain.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py tests/test_gpu_statistical.py tests/test_gpu_fullsize.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -3
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --chain 3000 --warmup 0 --steps 100 > gpurun_out/r06_bench_config4_chain_last.json 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --mt-prior sparse > gpurun_out/r06_bench_config4_sparse_last.json 2> /dev/null
for f in config4_chain config4_sparse; do python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_${f}_last.json').read().strip().splitlines()[-1]); ch=d.get('chain') or {}
print('$f', round(d['value'],2), round(d['ms_per_step'],3), 'dev', round(d['config']['device_sweep_ms'],3), 'launch', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3), ch.get('chain_total_s'), [round(x,2) for x in ch.get('window_mean_ms', [])][-8:])"; done
