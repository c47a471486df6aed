#!/bin/bash
# A/B of the resident-sampler sweep against the launch-per-block sweep on the bench workloads (gpurun_out/r04_ab/).
# usage: scripts/r04_ab.sh [tests] [workload ...]   (tests: run the GPU parity file with the mode forced on first)
out=gpurun_out/r04_ab; mkdir -p $out
if [ "$1" = "tests" ]; then
  shift
  JWAS_HIP_RESIDENT=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $out/parity_resident.log 2>&1
  tail -5 $out/parity_resident.log
fi
run() { name=$1; shift
  for m in 0 1; do
    JWAS_HIP_RESIDENT=$m timeout 600 python bench.py --no-cpu-baseline --via-api 0 "$@" > $out/${name}_res$m.json 2> $out/${name}_res$m.err
    grep -i -A1 'gave up' $out/${name}_res$m.err | cut -c1-600
    echo "$name resident=$m: $(python - <<PY
import json
try:
    d=json.loads(open('$out/${name}_res$m.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'], d['roofline']['frac'])
except Exception as e: print('FAILED', e)
PY
)"
  done; }
want=${@:-config3 pifixed config4 refbench packed config2}
for w in $want; do case $w in
  config3) run config3 --workload config3;;
  pifixed) run pifixed --workload config2 --pi-fixed 0.95;;
  config4) run config4 --workload config4;;
  refbench) run refbench --workload refbench;;
  packed) run packed --workload config2 --storage packed2bit;;
  config2) run config2 --workload config2;;
esac; done
