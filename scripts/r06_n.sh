#!/bin/bash
# Round 6: config 3's steady state (4 x 1024 markers per launch, 2-5 000 changes per sweep) with one sampler workgroup per block?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_n; mkdir -p $OUT
B="--no-cpu-baseline --via-api 0"
for pp in default 1; do
  if [ $pp = 1 ]; then export JWAS_HIP_PINGPONG=1; else unset JWAS_HIP_PINGPONG; fi
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py $B --workload config3 --burnin 1700 --warmup 0 --steps 20 > $OUT/bench_config3_long_pp$pp.json 2> $OUT/bench_config3_long_pp$pp.log
  grep "jwas_hip\] blocks" $OUT/bench_config3_long_pp$pp.log | tail -1 | grep -o "blocks=[0-9]* events=[0-9]*\|role=[0-9]*\|group:.*"
done
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d frac=%.3f launch_us=%.2f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
PY
done
