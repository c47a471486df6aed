#!/bin/bash
# A/B of the compact chain (sampler_st.hpp: compact_walk) against the speculative rounds (JWAS_HIP_COMPACT_OFF=1) on the
# sampler-bound workloads (gpurun_out/r04_compact/).  usage: scripts/r04_compact.sh [tests] [workload ...]
out=gpurun_out/r04_compact; mkdir -p $out
if [ "$1" = "tests" ]; then
  shift
  timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
  tail -5 $out/gpu_tests.log
fi
run() { name=$1; shift
  for m in off on; do
    if [ $m = off ]; then export JWAS_HIP_COMPACT_OFF=1; else unset JWAS_HIP_COMPACT_OFF; fi
    JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 "$@" > $out/${name}_$m.json 2> $out/${name}_$m.err
    echo "$name compact=$m: $(python - <<PY
import json
try:
    d=json.loads(open('$out/${name}_$m.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'], d['roofline']['frac'])
except Exception as e: print('FAILED', e)
PY
)"
    grep "jwas_hip\] blocks" $out/${name}_$m.err | tail -1 | sed 's/stage: assign.*resident=[01]//' | cut -c1-400
  done; unset JWAS_HIP_COMPACT_OFF; }
want=${@:-config3 pifixed config2}
for w in $want; do case $w in
  config3) run config3 --workload config3;;
  pifixed) run pifixed --workload config2 --pi-fixed 0.95;;
  config4s) run config4s --workload config4 --mt-prior sparse;;
  packed) run packed --workload config2 --storage packed2bit;;
  config2) run config2 --workload config2;;
esac; done
