#!/bin/bash
# Round 6, first GPU call: the new literal-chain tests at the production geometries, the t = 4 Rule T guard,
# and the two BASELINE configs whose bench lines were transients (item 4): config 3 and config 4 on their whole chains.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_literal.py tests/test_gpu_rule_t.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -25 > $OUT/gpu_tests.log
tail -25 $OUT/gpu_tests.log
B="--no-cpu-baseline --via-api 0"
export JWAS_HIP_DEBUG_PHASES=0
timeout 600 python bench.py $B --workload config3 --chain 1500 --warmup 0 --steps 100 > $OUT/bench_config3_chain.json 2> $OUT/bench_config3_chain.log
timeout 600 python bench.py $B --workload config4 --chain 3000 --warmup 0 --steps 100 > $OUT/bench_config4_chain.json 2> $OUT/bench_config4_chain.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"]))
    ch=d["chain"]; print(" chain_total_s=%.1f worst=%.1f@%d" % (ch["chain_total_s"], ch["worst_sweep_ms"], ch["worst_sweep_index"]))
    for k in ("window_mean_ms","window_events_per_sweep","window_markers_in_model","window_block_size"): print("  ",k,[round(v,1) for v in ch[k]])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
