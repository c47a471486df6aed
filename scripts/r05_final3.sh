#!/bin/bash
# Round 5, FINAL code: full GPU suite (parallel), profiles of the default (dense, grouped) and of the packed default (grouped), A/B lines on one box
cd $GRAFT_REPO_ROOT
set -u
TAG=r05k
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -6 > "$OUT/gpu_tests.log"
cat "$OUT/gpu_tests.log"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
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
run config2_groups4 30 --groups 4
run config2_packed 10 --storage packed2bit
python bench.py --no-cpu-baseline --via-api 0 --storage packed2bit --groups 0 > "$OUT/bench_config2_packed_groups0.json" 2>/dev/null
python bench.py --no-cpu-baseline --via-api 0 --groups 0 > "$OUT/bench_config2_groups0.json" 2>/dev/null
for f in "$OUT"/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],2), d['config']['blocks_per_launch'], round(d['config']['setup_s'],1), d.get('via_api',{}).get('value'))"; done
