#!/bin/bash
# Round 5, final code: profiles of the headline (grouped 4 / 2 / one block per launch), the full GPU suite, a long fuzz of the grouped schedule
cd $GRAFT_REPO_ROOT
set -u
TAG=r05h
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
run() {   # name, timed steps, bench args...
    local W=$1 K=$2; shift 2
    mkdir -p "$OUT/$W"
    python bench.py --via-api 0 --no-cpu-baseline "$@" > "$OUT/bench_$W.json" 2> "$OUT/$W/bench.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/ktrace" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps $K "$@" > "$OUT/$W/bench_under_rocprof.json" 2> "$OUT/$W/ktrace.err"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$W/pmc_fetch" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_fetch.json" 2> "$OUT/$W/pmc_fetch.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$W/pmc_write" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_write.json" 2> "$OUT/$W/pmc_write.err"
    python scripts/summarize_profiles.py "$OUT/$W" $K 3 "$W"
    find "$OUT/$W" -name "*kernel_trace.csv" -size +5M -delete
    find "$OUT/$W" -name "*counter_collection.csv" -size +5M -delete
    find "$OUT/$W" -name "*.db" -delete
}
run config2_groups4 30 --groups 4
run config2_groups2 30 --groups 2
run config2 30 --groups 0
python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -6 > "$OUT/gpu_tests.log"
cat "$OUT/gpu_tests.log"
( time JWAS_FUZZ_CASES=4000 timeout 1500 python -m pytest tests/test_gpu_groups.py -q -n 8 -k "random" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) > "$OUT/fuzz_grouped_1000_cases.log" 2>&1
cat "$OUT/fuzz_grouped_1000_cases.log"
grep -h "k_cross_mfma128\|k_gram_mfma" "$OUT"/config2_groups4/kernel_stats.csv "$OUT"/config2/kernel_stats.csv | cut -c1-200
