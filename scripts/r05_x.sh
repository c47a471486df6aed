#!/bin/bash
# Round 5, final code: config5shard with the default (grouped) schedule -- bench line, kernel trace, PMC
cd $GRAFT_REPO_ROOT
set -u
TAG=r05i
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
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
run config5shard 10 --workload config5shard
python bench.py --no-cpu-baseline --via-api 0 --workload config5shard --groups 0 > "$OUT/bench_config5shard_plain.json" 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -k "loopback or sharded_sweep_single or full_size_config2" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
for f in "$OUT"/bench_config5shard*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['blocks_per_launch'], d['config']['setup_s'])"; done
