#!/bin/bash
# Round 6: the policy e2e tests, the (4, 512) literal case, smoke(), and config 3's chain with the final code.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_o; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_literal.py -m gpu -q -n 4 -k "pingpong or literal" 2>&1 | grep -v "$F" | tail -8 > $OUT/gpu_tests.log; tail -5 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --no-cpu-baseline --via-api 0 --workload config3 --chain 1500 --warmup 0 --steps 100 > $OUT/bench_config3_chain.json 2> /dev/null
python - $OUT/bench_config3_chain.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
print("it/s=%.2f ms=%.2f events=%.0f bs=%d m=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"]))
ch=d["chain"]; print(" chain_total_s=%.1f" % ch["chain_total_s"]); 
for k in ("window_mean_ms","window_events_per_sweep","window_block_size"): print("  ",k,[round(v,1) for v in ch[k]])
PY
