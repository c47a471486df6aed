#!/bin/bash
# development build: role-split timing of the dense sweep (results of ROLE=1/2 runs are wrong by design)
mkdir -p gpurun_out/r03d
for role in 0 1 2; do
for bs in 512; do
  JWAS_HIP_DEBUG_ROLE=$role timeout 600 python bench.py --workload refbench --steps 5 --warmup 5 --burnin 0 --no-cpu-baseline --block-size $bs > gpurun_out/r03d/refbench_${bs}_$role.json 2> gpurun_out/r03d/refbench_${bs}_$role.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03d/refbench_${bs}_$role.json"))
print("role=$role bs=$bs", "sweep_ms", round(d["config"]["device_sweep_ms"],2), "events", d["config"]["events_per_sweep"])
PY
done
done
