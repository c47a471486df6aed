#!/bin/bash
# Round 6, final code: the whole GPU suite (default + ping-pong forced on for the grouped files), the workloads whose schedule changed
# after the profile run (fixed pi on 4-block ping-pong, packed), fuzz.
cd $GRAFT_REPO_ROOT
set -u
TAG=r06b
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -6 > "$OUT/gpu_tests.log"; tail -3 "$OUT/gpu_tests.log"
JWAS_HIP_PINGPONG=1 timeout 1200 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py tests/test_gpu_e2e.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -4 > "$OUT/gpu_tests_pingpong_forced.log"; tail -2 "$OUT/gpu_tests_pingpong_forced.log"
run() {   # name, timed steps, bench args...
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
run config2_pifixed 10 --pi-fixed 0.95 --warmup 20
run config2_packed 10 --storage packed2bit
run config3 10 --workload config3 --warmup 20
JWAS_BENCH_GROUPS_SMALL=4 python bench.py --no-cpu-baseline --via-api 0 --workload config3 --steps 30 > "$OUT/bench_config3_pingpong4.json" 2> /dev/null
python bench.py > "$OUT/bench_default.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --chain 3000 --warmup 0 --steps 100 > "$OUT/bench_config4_chain.json" 2> /dev/null
( time JWAS_FUZZ_CASES=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_groups.py -q -n 8 -k "random" 2>&1 | grep -v "$F" | tail -5 ) > "$OUT/fuzz_6000_cases.log" 2>&1
tail -4 "$OUT/fuzz_6000_cases.log"
