#!/bin/bash
# Round 5: Rule T with exceptions: parity tests, fuzz, then the config-4 chain (150 sweeps, solved / fallen / exceptions per sweep).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_rule_t.py -x -q 2>&1 | tail -25 > $OUT/rule_t_tests.log; cat $OUT/rule_t_tests.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -n 4 -k rule_t 2>&1 | tail -12 > $OUT/fuzz.log; cat $OUT/fuzz.log
for m in BayesC BayesB; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 --steps ${STEPS:-150} --workload config4 --warmup 0 --burnin 0 --mt-method $m > $OUT/chain_$m.json 2> $OUT/chain_$m.log
  grep "jwas_hip\] blocks" $OUT/chain_$m.log | sed 's/.*compact: blocks=\([0-9]*\) fallback=\([0-9]*\).*xchain=\([0-9]*\).*/\1 \2 \3/' | awk 'NR%10==0 {printf "%d:%s/%s/%s ", NR, $1, $2, $3} END {print ""}' > $OUT/solved_$m.txt
  echo "== $m (sweep:solved/fallen/exceptions)"; cat $OUT/solved_$m.txt
  python - $OUT/chain_$m.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); print("%d steps: ms=%.2f sweep=%.2f" % (d["steps"], d["ms_per_step"], d["config"]["device_sweep_ms"]))
PY
  grep "jwas_hip\] blocks" $OUT/chain_$m.log | tail -1 | cut -c1-500
done
