#!/bin/bash
# Round-6 measurement artefacts on the GPU box:  bash scripts/profile_round5.sh r06   (then: python scripts/collect_profiles.py r06 r06)
# For EVERY workload bench.py offers: the plain bench line (with its CPU baseline), rocprofv3 kernel trace + stats of the same
# command, FETCH_SIZE / WRITE_SIZE in separate --pmc passes (counters only -- never combined with a trace).
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -6 > "$OUT/gpu_tests.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
run() {   # name, timed steps, bench args...
    local W=$1 K=$2; shift 2
    mkdir -p "$OUT/$W"
    python bench.py --via-api 0 "$@" > "$OUT/bench_$W.json" 2> "$OUT/$W/bench.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/ktrace" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps $K "$@" > "$OUT/$W/bench_under_rocprof.json" 2> "$OUT/$W/ktrace.err"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$W/pmc_fetch" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_fetch.json" 2> "$OUT/$W/pmc_fetch.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$W/pmc_write" -o bench -- python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" > "$OUT/$W/bench_under_pmc_write.json" 2> "$OUT/$W/pmc_write.err"
    python scripts/summarize_profiles.py "$OUT/$W" $K 3 "$W"
    # the raw traces are large: keep only the summaries
    find "$OUT/$W" -name "*kernel_trace.csv" -size +5M -delete
    find "$OUT/$W" -name "*counter_collection.csv" -size +5M -delete
    find "$OUT/$W" -name "*.db" -delete
}
run config2 30
run config3 10 --workload config3 --warmup 20
# config 3 on its steady state: 1500 chain sweeps before the timed region (BayesR sheds markers for > 1000 sweeps)
run config3_longrun 10 --workload config3 --burnin 1500 --warmup 0
run config2_pifixed 10 --pi-fixed 0.95 --warmup 20
run refbench 10 --workload refbench --warmup 10 --burnin 0
run config4 10 --workload config4 --warmup 10 --burnin 0
run config4_walk 10 --workload config4 --warmup 10 --burnin 0 --no-section-solve
run config4_sparse 10 --workload config4 --mt-prior sparse
run config4_bayesb 10 --workload config4 --mt-method BayesB --warmup 10 --burnin 0
run config5shard 10 --workload config5shard
run config2_packed 10 --storage packed2bit
# the long differential fuzz (device vs oracle bit for bit; JWAS_FUZZ_CASES/8 of them are Rule T cases)
( time JWAS_FUZZ_CASES=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 -k "random" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 ) > "$OUT/fuzz_6000_cases.log" 2>&1
# single-trait helper workgroup A/B on the reference's own benchmark
JWAS_HIP_CORR_HELPER=0 python bench.py --no-cpu-baseline --via-api 0 --workload refbench --warmup 10 --burnin 0 > "$OUT/bench_refbench_nohelper.json" 2> /dev/null
# the two BASELINE configs whose chains pass through several regimes, every sweep from the start on the clock (bench.py --chain):
# chain total, worst sweep, 100-sweep window means + the steady-state timed region behind it
python bench.py --no-cpu-baseline --via-api 0 --workload config3 --chain 1500 --warmup 0 --steps 100 > "$OUT/bench_config3_chain.json" 2> /dev/null
python bench.py --no-cpu-baseline --via-api 0 --workload config4 --chain 3000 --warmup 0 --steps 100 > "$OUT/bench_config4_chain.json" 2> /dev/null
# one rank's share of config 3 INSIDE the sharded full chain (8 shards emulated as 8 contexts: each shard's device sweep time with
# its real share of the chain's turnover), ping-pong pairs on
python scripts/shard_check.py --method BayesR --shards 8 --iters 90 --time-shards --pairs --json "$OUT/rank_share_config3_inchain.json" > "$OUT/rank_share_config3_inchain.log" 2>&1
ls "$OUT"
