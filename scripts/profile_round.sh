#!/bin/bash
# Collect the round's measurement artefacts on the GPU box:  bash scripts/profile_round.sh r02
# For the headline workload (python bench.py) and the two sampler-bound ones (config3, config2 --pi-fixed 0.95):
#   plain bench line, rocprofv3 kernel trace + stats of the same command, FETCH_SIZE / WRITE_SIZE in separate --pmc passes
#   (counters only -- never combined with a trace).  Outputs land in gpurun_out/<tag>/<workload>/;
#   scripts/summarize_profiles.py condenses them into the files kept under profiles/ (+ profiles/traffic.json rows).
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ONLY=${2:-all}          # "dense": only the dense-regime workloads (refbench, config4) -- bash scripts/profile_round.sh r02 dense
if [ "$ONLY" = all ]; then
python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > "$OUT/gpu_tests.log"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
fi
run() {   # name, timed steps, bench args...
    local W=$1 K=$2; shift 2
    mkdir -p "$OUT/$W"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/ktrace" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps $K "$@" > "$OUT/$W/bench_under_rocprof.json" 2> "$OUT/$W/ktrace.err"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$W/pmc_fetch" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_fetch.json" 2> "$OUT/$W/pmc_fetch.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$W/pmc_write" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_write.json" 2> "$OUT/$W/pmc_write.err"
    python scripts/summarize_profiles.py "$OUT/$W" $K 3 "$W"
    # the raw traces are large: keep only the summaries
    find "$OUT/$W" -name "*kernel_trace.csv" -size +5M -delete
    find "$OUT/$W" -name "*counter_collection.csv" -size +5M -delete
    find "$OUT/$W" -name "*.db" -delete
}
if [ "$ONLY" = all ]; then
run config2 30
run config3 10 --workload config3 --warmup 20
run config2_pifixed 10 --pi-fixed 0.95 --warmup 20
else
run refbench 10 --workload refbench --warmup 10 --burnin 0
run config4 10 --workload config4 --warmup 10 --burnin 0
fi
ls -la "$OUT" "$OUT"/*
