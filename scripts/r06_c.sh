#!/bin/bash
# Round 6, third GPU call: the hoisted literal evaluation (mt1_eval_hoisted) in the dense multi-trait walks -- parity (multi-trait
# parity tests, Rule T tests, the differential fuzz: bit for bit against the oracle) and config 4's whole chain.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py tests/test_gpu_fuzz.py -m gpu -q -n 4 -k "mt or multitrait or rule_t or fuzz or random or bit" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -12 > $OUT/gpu_tests.log
tail -12 $OUT/gpu_tests.log
B="--no-cpu-baseline --via-api 0"
timeout 600 python bench.py $B --workload config4 --chain 3000 --warmup 0 --steps 100 > $OUT/bench_config4_chain.json 2> $OUT/bench_config4_chain.log
timeout 300 python bench.py $B --workload config4 --warmup 10 --burnin 0 --steps 30 > $OUT/bench_config4.json 2> $OUT/bench_config4.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"]))
    ch=d.get("chain")
    if ch:
        print(" chain_total_s=%.1f worst=%.1f@%d" % (ch["chain_total_s"], ch["worst_sweep_ms"], ch["worst_sweep_index"]))
        for k in ("window_mean_ms","window_events_per_sweep","window_block_size"): print("  ",k,[round(v,1) for v in ch[k]])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
