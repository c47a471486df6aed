#!/bin/bash
# Round 6: three-level block policy of a dense-start multi-trait chain (256 / 512 / 1024): tests + config 4's whole chain.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "three_resident or skip_and_verify or switching" 2>&1 | grep -v "$F" | tail -15
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_statistical.py tests/test_gpu_fullsize.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -4
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --chain 3000 --warmup 0 --steps 100 > gpurun_out/r06_bench_config4_chain_1024.json 2> /dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_config4_chain_1024.json').read().strip().splitlines()[-1]); ch=d['chain']
print('config4 chain', round(ch['chain_total_s'],2), 's; steady', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms dev', round(d['config']['device_sweep_ms'],3), 'launch', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3))
print([round(x,1) for x in ch['window_mean_ms']]); print(ch['window_block_size'])"
