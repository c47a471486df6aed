#!/bin/bash
# scratch: GPU experiments of the moment (not part of the measurement set)
for w in "config4"; do
  echo "== $w"
  JWAS_HIP_DEBUG_PHASES=1 timeout 900 python bench.py --workload $w --steps 3 --warmup 2 --burnin 30 --no-cpu-baseline 2>&1 | grep -E "jwas_hip\]" | tail -1 | cut -c1-420
done
