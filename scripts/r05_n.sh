#!/bin/bash
# Round 5: phase counters of the multi-trait SPARSE regime (config 4's steady state): 512- vs 256-marker blocks
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_n; mkdir -p $OUT
for bs in 512 256 1024; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py --no-cpu-baseline --via-api 0 --workload config4 --mt-prior sparse --block-size $bs > $OUT/sparse_$bs.json 2> $OUT/sparse_$bs.log
  echo "== bs $bs"; grep "jwas_hip\] blocks" $OUT/sparse_$bs.log | tail -1 | cut -c1-330
  python - $OUT/sparse_$bs.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); print("ms=%.2f sweep=%.2f launch_us=%.2f" % (d["ms_per_step"], d["config"]["device_sweep_ms"], d["roofline"]["avg_launch_us"]))
except Exception as e: print("ERR", e)
PY
done
