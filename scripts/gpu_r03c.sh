#!/bin/bash
JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --workload refbench --steps 5 --warmup 5 --burnin 0 --no-cpu-baseline 2>&1 >/dev/null | tail -1
