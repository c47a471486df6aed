#!/bin/bash
# Round 6, second GPU call: phase counters of the sampler-bound regimes (baseline of the round) and the block-size threshold of the
# adaptive policy along config 3's chain (BayesR sheds markers for > 1000 sweeps: where should 512 -> 1024 happen?).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_literal.py tests/test_gpu_rule_t.py -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -8 > $OUT/gpu_tests.log
tail -8 $OUT/gpu_tests.log
B="--no-cpu-baseline --via-api 0"
export JWAS_HIP_DEBUG_PHASES=1
timeout 300 python bench.py $B --workload config4 --mt-prior sparse --steps 10 > $OUT/bench_config4_sparse.json 2> $OUT/bench_config4_sparse.log
timeout 300 python bench.py $B --workload config3 --steps 10 > $OUT/bench_config3.json 2> $OUT/bench_config3.log
timeout 300 python bench.py $B --workload config2 --pi-fixed 0.95 --steps 10 > $OUT/bench_pifixed.json 2> $OUT/bench_pifixed.log
unset JWAS_HIP_DEBUG_PHASES
for w in config4_sparse config3 pifixed; do echo "== $w"; grep "jwas_hip\] blocks" $OUT/bench_$w.log | tail -1 | cut -c1-700; done
for fr in 0.025 0.05; do
JWAS_ADAPTIVE_FRACTION=$fr timeout 600 python bench.py $B --workload config3 --chain 700 --warmup 0 --steps 10 > $OUT/bench_config3_chain_$fr.json 2> $OUT/bench_config3_chain_$fr.log
done
JWAS_ADAPTIVE_FRACTION=0.05 timeout 600 python bench.py $B --workload config3 --groups 0 --chain 700 --warmup 0 --steps 10 > $OUT/bench_config3_chain_0.05_g0.json 2> $OUT/bench_config3_chain_g0.log
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
