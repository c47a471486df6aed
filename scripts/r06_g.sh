#!/bin/bash
# Round 6: the host policy with ping-pong pairs (bench.py / mcmc.run_chain defaults), e2e chains, statistical + full-size smoke.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_g; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_groups.py tests/test_gpu_literal.py -m gpu -q -n 4 2>&1 | grep -v "$F" | tail -8 > $OUT/gpu_tests.log
tail -8 $OUT/gpu_tests.log
B="--no-cpu-baseline --via-api 0"
timeout 300 python bench.py $B --workload config3 --steps 30 > $OUT/bench_config3.json 2> $OUT/bench_config3.log
timeout 300 python bench.py $B --workload config2 --pi-fixed 0.95 --steps 30 > $OUT/bench_pifixed.json 2> $OUT/bench_pifixed.log
timeout 600 python bench.py $B --workload config3 --chain 1500 --warmup 0 --steps 100 > $OUT/bench_config3_chain.json 2> $OUT/bench_config3_chain.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d frac=%.3f gsetup=%.1f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"], c.get("group_setup_s",0)))
    ch=d.get("chain")
    if ch:
        print(" chain_total_s=%.1f worst=%.1f@%d" % (ch["chain_total_s"], ch["worst_sweep_ms"], ch["worst_sweep_index"]))
        for k in ("window_mean_ms","window_events_per_sweep","window_block_size"): print("  ",k,[round(v,1) for v in ch[k]])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
