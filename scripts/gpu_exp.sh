#!/bin/bash
# scratch: GPU experiments of the moment (not part of the measurement set).  Phase cycle counts of the sampler role:
#   JWAS_HIP_DEBUG_PHASES=1 python bench.py --workload config3 --steps 3 --warmup 2 --no-cpu-baseline
for w in "config2" "config3" "config4" "refbench"; do
  echo "== $w"
  JWAS_HIP_DEBUG_PHASES=1 timeout 900 python bench.py --workload $w --steps 3 --warmup 2 --burnin 30 --no-cpu-baseline 2>&1 | grep -E "jwas_hip\]" | tail -1 | cut -c1-330
done
