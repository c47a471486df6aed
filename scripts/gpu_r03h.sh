#!/bin/bash
for w in "config4 --mt-prior sparse" "config3" "config2 --pi-fixed 0.95"; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --via-api 0 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'it/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'sweep_ms', round(d['config']['device_sweep_ms'],2), 'bs', d['config']['block_size'], 'frac', round(d['roofline']['frac'],3), 'events', round(d['config']['events_per_sweep']))
"
done
