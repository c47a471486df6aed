#!/bin/bash
# Round 5: grouped launches (k_group_step) -- parity tests, then the headline with 1 / 2 / 4 blocks per launch (dense and packed).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_p; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_groups.py -x -q 2>&1 | tail -15 > $OUT/tests.log
cat $OUT/tests.log
for g in 0 2 4 0 2 4; do
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 --groups $g 2>$OUT/err_$g.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense groups=$g ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f setup=%.1f events=%.0f' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['setup_s'], d['config']['events_per_sweep']))" | tee -a $OUT/bench.log
done
for g in 0 2 4; do
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 --storage packed2bit --groups $g 2>$OUT/errp_$g.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('packed groups=$g ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" | tee -a $OUT/bench.log
done
tail -5 $OUT/err_4.log
