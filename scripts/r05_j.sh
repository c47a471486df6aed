#!/bin/bash
# Round 5: cycles of the first verification pass | the exception passes (counters 13 / 14) along a config-4 chain
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_j; mkdir -p $OUT
for m in BayesC; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 --steps ${STEPS:-300} --workload config4 --warmup 0 --burnin 0 --mt-method $m > $OUT/chain_$m.json 2> $OUT/chain_$m.log
  grep "jwas_hip\] blocks" $OUT/chain_$m.log | sed 's/.*update wg0: share=\([0-9]*\) wait=\([0-9]*\).*compact: blocks=\([0-9]*\) fallback=\([0-9]*\).*tailwait=\([0-9]*\).*xchain=\([0-9]*\).*/\3 \4 \6 \1 \2 \5/' | awk 'NR%25==0 {printf "%d: solved=%s fallen=%s exc=%s pass1=%s passN=%s window=%s\n", NR, $1, $2, $3, $4, $5, $6}'
  python - $OUT/chain_$m.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); print("%d steps: ms=%.2f sweep=%.2f" % (d["steps"], d["ms_per_step"], d["config"]["device_sweep_ms"]))
PY
done
