#!/bin/bash
# Round 5: a LONG config-4 chain (1000 sweeps): where do the exceptions per sweep settle, and the sweep time with them
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_i; mkdir -p $OUT
for m in BayesC BayesB; do
  for v in solve walk; do
    X=""; [ $v = walk ] && X="--no-section-solve"
    JWAS_HIP_DEBUG_PHASES=1 timeout 900 python bench.py --no-cpu-baseline --via-api 0 --steps 200 --workload config4 --warmup 0 --burnin 800 --mt-method $m $X > $OUT/chain_${m}_$v.json 2> $OUT/chain_${m}_$v.log
    grep "jwas_hip\] blocks" $OUT/chain_${m}_$v.log | sed 's/.*compact: blocks=\([0-9]*\) fallback=\([0-9]*\).*xchain=\([0-9]*\).*/\1 \2 \3/' | awk 'NR%50==0 {printf "%d:%s/%s/%s ", NR, $1, $2, $3} END {print ""}' > $OUT/solved_${m}_$v.txt
    echo "== $m $v (sweep:solved/fallen/exceptions)"; cat $OUT/solved_${m}_$v.txt
    python - $OUT/chain_${m}_$v.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); print("sweeps 801..1000: ms=%.2f sweep=%.2f it/s=%.1f" % (d["ms_per_step"], d["config"]["device_sweep_ms"], d["value"]))
PY
  done
done
