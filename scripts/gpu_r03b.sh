#!/bin/bash
# dense_big_st: parity + refbench at block 512 / 256 / 128
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -q -x -k "dense_big or cooperative_dense_apply_is or device_resident or all_included" 2>&1 | tail -15 > gpurun_out/r03b/tests.log
cat gpurun_out/r03b/tests.log | tail -6
for bs in 512 256 128; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --workload refbench --steps 10 --warmup 10 --burnin 0 --no-cpu-baseline --block-size $bs > gpurun_out/r03b/refbench_$bs.json 2> gpurun_out/r03b/refbench_$bs.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03b/refbench_$bs.json"))
print("bs=$bs", "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],2), "sweep_ms", round(d["config"]["device_sweep_ms"],2), "frac", round(d["roofline"]["frac"],3), "events", d["config"]["events_per_sweep"])
PY
  tail -2 gpurun_out/r03b/refbench_$bs.err
done
