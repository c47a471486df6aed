#!/bin/bash
# Round 6: multi-trait chains that START sparse take 1024-marker blocks below 0.5 % turnover too (run_chain / bench.py --mt-prior sparse):
# e2e + host-policy GPU tests, and the profile of that line (kernel trace, FETCH_SIZE / WRITE_SIZE passes).
cd $GRAFT_REPO_ROOT
set -u
TAG=r06g
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_statistical.py tests/test_gpu_fullsize.py tests/test_bench_contract.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -4
run() {
    local W=$1 K=$2; shift 2
    mkdir -p "$OUT/$W"
    python bench.py --via-api 0 "$@" > "$OUT/bench_$W.json" 2> "$OUT/$W/bench.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/ktrace" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps $K "$@" > "$OUT/$W/bench_under_rocprof.json" 2> "$OUT/$W/ktrace.err"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$W/pmc_fetch" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_fetch.json" 2> "$OUT/$W/pmc_fetch.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$W/pmc_write" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_write.json" 2> "$OUT/$W/pmc_write.err"
    python scripts/summarize_profiles.py "$OUT/$W" $K 3 "$W"
    find "$OUT/$W" -name "*kernel_trace.csv" -size +5M -delete
    find "$OUT/$W" -name "*counter_collection.csv" -size +5M -delete
    find "$OUT/$W" -name "*.db" -delete
}
run config4_sparse 10 --workload config4 --mt-prior sparse --no-cpu-baseline
python -c "
import json; d=json.loads(open('$OUT/bench_config4_sparse.json').read().strip().splitlines()[-1])
print('config4_sparse', round(d['value'],2), round(d['ms_per_step'],3), 'dev', round(d['config']['device_sweep_ms'],3), 'launch', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3), d['config']['block_size'], d['config']['block_policy'], d['roofline'].get('traffic'))"
