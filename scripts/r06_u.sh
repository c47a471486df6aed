#!/bin/bash
# Round 6, final code (late hand-over + cooperative apply in grouped launches): GPU suite, forced ping-pong, grouped fuzz, profiles.
cd $GRAFT_REPO_ROOT
set -u
TAG=r06d
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
run config3 10 --workload config3 --warmup 20
run config2_pifixed 10 --pi-fixed 0.95 --warmup 20
JWAS_BENCH_GROUPS_SMALL=2 python bench.py --no-cpu-baseline --via-api 0 --workload config3 --steps 30 > "$OUT/bench_config3_pairs.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config3 --chain 1500 --warmup 0 --steps 100 > "$OUT/bench_config3_chain.json" 2> /dev/null
python bench.py > "$OUT/bench_default.json" 2> /dev/null
( time JWAS_FUZZ_CASES=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_groups.py -q -n 8 -k "random" 2>&1 | grep -v "$F" | tail -5 ) > "$OUT/fuzz_cases.log" 2>&1
tail -4 "$OUT/fuzz_cases.log"
