#!/bin/bash
for role in 0 3; do
for bs in 512 128; do
  JWAS_HIP_DEBUG_PHASES=1 JWAS_HIP_DEBUG_ROLE=$role timeout 600 python bench.py --workload refbench --steps 5 --warmup 5 --burnin 0 --no-cpu-baseline --block-size $bs 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('role=$role bs=$bs', 'sweep_ms', round(d['config']['device_sweep_ms'],2), 'events', d['config']['events_per_sweep'])
"
tail -1 /tmp/e.txt | sed 's/.*update wg0/update wg0/'
done
done
