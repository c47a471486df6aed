#!/bin/bash
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "dense_big" 2>&1 | tail -3
for bs in 512 256; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --workload refbench --steps 5 --warmup 5 --burnin 0 --no-cpu-baseline --block-size $bs > gpurun_out/r03c/refbench_$bs.json 2> gpurun_out/r03c/refbench_$bs.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c/refbench_$bs.json"))
print("bs=$bs", "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],2), "sweep_ms", round(d["config"]["device_sweep_ms"],2), "frac", round(d["roofline"]["frac"],3), "events", d["config"]["events_per_sweep"])
PY
  tail -1 gpurun_out/r03c/refbench_$bs.err
done
