#!/bin/bash
# Round 5: the 128 x 128-tile cross-Gram kernel of grouped launches (set-up time), tests on the final code
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_u; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_groups.py -x -q -k "mfma or literal or restatement" 2>&1 | tail -8 | tee $OUT/tests.log
for g in 4 2 0; do
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 --groups $g 2>$OUT/err_$g.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense groups=$g ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f setup=%.1f events=%.0f' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['setup_s'], d['config']['events_per_sweep']))" | tee -a $OUT/bench.log
done
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -k "full_size_config2 or sharded_sweep_single" 2>&1 | tail -5 | tee -a $OUT/tests.log
