#!/bin/bash
# Round 5: (1) how the share of solved sections evolves along a config-4 chain (pi estimated), (2) the ST kernels after the
# compiler barriers around stage_load's younger loads: parity subset + config 3 / headline A/B, (3) Float64 weights, full-size Rule T test.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_f64.py tests/test_gpu_fullsize.py -x -q -k "weights or rule_t" 2>&1 | tail -8 > $OUT/tests_new.log; cat $OUT/tests_new.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -n 4 2>&1 | tail -4 > $OUT/tests_parity.log; cat $OUT/tests_parity.log
for m in BayesC BayesB; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 --steps 150 --workload config4 --warmup 0 --burnin 0 --mt-method $m > $OUT/chain_$m.json 2> $OUT/chain_$m.log
  grep "jwas_hip\] blocks" $OUT/chain_$m.log | sed 's/.*compact: blocks=\([0-9]*\) fallback=\([0-9]*\).*/\1 \2/' | awk '{printf "%d:%s/%s ", NR, $1, $2} END {print ""}' > $OUT/solved_$m.txt
  echo "== $m (sweep:solved/fallen)"; cat $OUT/solved_$m.txt | cut -c1-1500
  python - $OUT/chain_$m.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); print("150 steps: ms=%.2f sweep=%.2f" % (d["ms_per_step"], d["config"]["device_sweep_ms"]))
PY
done
python bench.py --no-cpu-baseline --via-api 0 --workload config3 --warmup 20 > $OUT/bench_config3.json 2>/dev/null
python bench.py --no-cpu-baseline --via-api 0 > $OUT/bench_config2.json 2>/dev/null
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f" % (d["value"], d["ms_per_step"]))
PY
done
