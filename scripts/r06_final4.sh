#!/bin/bash
# Round 6, final code (PP split, skip and verify, two-step front): whole GPU suite, ping-pong forced, the four timed workloads with their
# kernel-trace / PMC profiles, the driver-style default bench, chains, fuzz.
cd $GRAFT_REPO_ROOT
set -u
TAG=r06e
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -6 > "$OUT/gpu_tests.log"; tail -3 "$OUT/gpu_tests.log"
JWAS_HIP_PINGPONG=1 timeout 1200 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py tests/test_gpu_e2e.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -4 > "$OUT/gpu_tests_pingpong_forced.log"; tail -2 "$OUT/gpu_tests_pingpong_forced.log"
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
python bench.py > "$OUT/bench_default.json" 2> /dev/null; cat "$OUT/bench_default.json" | cut -c1-400
run config2 10 --workload config2 --no-cpu-baseline
run config3 10 --workload config3 --warmup 20 --no-cpu-baseline
run config2_pifixed 10 --workload config2 --pi-fixed 0.95 --no-cpu-baseline
run config2_packed 10 --workload config2 --storage packed2bit --no-cpu-baseline
python bench.py --no-cpu-baseline --via-api 0 --workload config3 --burnin 1400 --warmup 0 --steps 30 > "$OUT/bench_config3_steady.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config3 --chain 1500 --warmup 0 --steps 100 > "$OUT/bench_config3_chain.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --chain 3000 --warmup 0 --steps 100 > "$OUT/bench_config4_chain.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --mt-prior sparse > "$OUT/bench_config4_sparse.json" 2> /dev/null
python bench.py --via-api 0 --workload config4 > "$OUT/bench_config4.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload refbench > "$OUT/bench_refbench.json" 2> /dev/null
for f in config4_sparse config4 refbench; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; done
for f in config3_steady config3_chain config4_chain; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config'].get('chain_total_s'), d['config'].get('worst_sweep_ms'))"; done
( time JWAS_FUZZ_CASES=8000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_groups.py -q -n 8 -k "random" 2>&1 | grep -v "$F" | tail -5 ) > "$OUT/fuzz_8000_cases.log" 2>&1
tail -4 "$OUT/fuzz_8000_cases.log"
