#!/bin/bash
# Round 5: grouped launches after the pair / four hierarchy fix -- parity + fuzz, then benches (setup time with the odd-only pairs)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_q; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_groups.py -x -q 2>&1 | tail -15 > $OUT/tests.log
cat $OUT/tests.log
for g in 4 2; do
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 --groups $g 2>$OUT/err_$g.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense groups=$g ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f setup=%.1f events=%.0f' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['setup_s'], d['config']['events_per_sweep']))" | tee -a $OUT/bench.log
done
for w in config3; do
for g in 0 4; do
  timeout 600 python bench.py --no-cpu-baseline --via-api 0 --workload $w --groups $g 2>$OUT/err3_$g.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w groups=$g ms=%.3f sweep=%.3f launch_us=%.2f frac=%.4f bs=%d m=%d' % (d['ms_per_step'], d['config']['device_sweep_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['block_size'], d['config']['blocks_per_launch']))" | tee -a $OUT/bench.log
done; done
