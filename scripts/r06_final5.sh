#!/bin/bash
# Round 6, the very last code (three-level multi-trait block policy, skip and verify on 1024-marker blocks): GPU suite, the new
# multi-trait differential fuzz, smoke, default bench, config 4's chain and sparse line.
cd $GRAFT_REPO_ROOT
set -u
OUT=$PWD/gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -6 > "$OUT/gpu_tests.log"; tail -3 "$OUT/gpu_tests.log"
( time JWAS_FUZZ_CASES=2400 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 -k "skip_and_verify or rule_t" 2>&1 | grep -v "$F" | tail -5 ) > "$OUT/fuzz_multitrait_600_cases.log" 2>&1; tail -5 "$OUT/fuzz_multitrait_600_cases.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -2
python bench.py > "$OUT/bench_default.json" 2> /dev/null; cut -c1-330 "$OUT/bench_default.json"
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --chain 3000 --warmup 0 --steps 100 > "$OUT/bench_config4_chain.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --mt-prior sparse > "$OUT/bench_config4_sparse.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --mt-prior sparse --block-size 1024 > "$OUT/bench_config4_sparse_1024.json" 2> /dev/null
for f in config4_chain config4_sparse config4_sparse_1024; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); ch=d.get('chain') or {}
print('$f', round(d['value'],2), round(d['ms_per_step'],3), 'dev', round(d['config']['device_sweep_ms'],3), 'launch', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3), ch.get('chain_total_s'), [round(x,1) for x in ch.get('window_mean_ms', [])][-14:], ch.get('window_max_ms', [0,0])[1:] and round(max(ch['window_max_ms'][1:]),1))"; done
