#!/bin/bash
# Round 5: Rule T (multi-trait): parity tests, then config 4 with the solve and with the walk (A/B on one box), phase counters.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_rule_t.py -x -q 2>&1 | tail -25 > $OUT/rule_t_tests.log; cat $OUT/rule_t_tests.log
B="--no-cpu-baseline --via-api 0 --steps 10 --workload config4 --warmup 5 --burnin 0"
export JWAS_HIP_DEBUG_PHASES=1
for v in solve walk; do
  X=""; [ $v = walk ] && X="--no-section-solve"
  timeout 300 python bench.py $B $X > $OUT/bench_config4_$v.json 2> $OUT/bench_config4_$v.log
  timeout 300 python bench.py $B --mt-method BayesB $X > $OUT/bench_config4_bayesb_$v.json 2> $OUT/bench_config4_bayesb_$v.log
done
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(".json", ".log")).read()[-1500:])
PY
done
for v in config4_solve config4_walk config4_bayesb_solve; do echo "== $v"; grep "jwas_hip\] blocks" $OUT/bench_$v.log | tail -1 | cut -c1-420; done
