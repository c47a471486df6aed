#!/bin/bash
# scratch: GPU experiments of the moment (not part of the measurement set)
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 2>&1 | tail -2
python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -2
for w in "config4" "config4 --mt-method BayesB"; do
  echo "== $w"
  JWAS_HIP_DEBUG_PHASES=1 timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --burnin 30 --no-cpu-baseline 2>&1 | grep -E "jwas_hip\]|ms_per_step" | tail -2 | cut -c1-330
done
