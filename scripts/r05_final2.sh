#!/bin/bash
# Round 5, final code: the full GPU suite exactly as the driver runs it (serial, -x), the packed workload's profile, default bench
cd $GRAFT_REPO_ROOT
set -u
TAG=r05j
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -6 ) > "$OUT/gpu_tests_serial.log" 2>&1
cat "$OUT/gpu_tests_serial.log"
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
run config2_packed 10 --storage packed2bit
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
for f in "$OUT"/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],2), d['config']['blocks_per_launch'], round(d['config']['setup_s'],1), d.get('via_api',{}).get('value'))"; done
python __graft_entry__.py smoke 2>&1 | tail -3
