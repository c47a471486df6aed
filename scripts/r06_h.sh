#!/bin/bash
# Round 6: dense multi-trait walks that skip the markers outside the model (parity), and where the chain should leave them.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_h; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py tests/test_gpu_fuzz.py -m gpu -q -n 4 -k "mt or multitrait or rule_t or fuzz or random or bit" 2>&1 | grep -v "$F" | tail -5 > $OUT/gpu_tests_mt.log
tail -5 $OUT/gpu_tests_mt.log
B="--no-cpu-baseline --via-api 0"
run() { local tag=$1; shift; env "$@" timeout 600 python bench.py $B --workload config4 --chain 1900 --warmup 0 --steps 20 > $OUT/bench_config4_chain_$tag.json 2> $OUT/bench_config4_chain_$tag.log; }
run a JWAS_X=0
run b JWAS_HIP_DENSE_MT_FRACTION=0.1 JWAS_MT_SPARSE_FRACTION=0.1
run c JWAS_HIP_DENSE_MT_FRACTION=0.03 JWAS_MT_SPARSE_FRACTION=0.03
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"]))
    ch=d.get("chain")
    if ch:
        print(" chain_total_s=%.1f worst=%.1f@%d" % (ch["chain_total_s"], ch["worst_sweep_ms"], ch["worst_sweep_index"]))
        for k in ("window_mean_ms","window_max_ms","window_events_per_sweep","window_block_size"): print("  ",k,[round(v,1) for v in ch[k]])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
