#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_j; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "row_shards" 2>&1 | grep -v "$F" | tail -40 > $OUT/gpu_tests_rows.log
tail -30 $OUT/gpu_tests_rows.log
B="--no-cpu-baseline --via-api 0"
JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B --storage packed2bit --steps 10 > $OUT/bench_packed.json 2> $OUT/bench_packed.log
grep "jwas_hip\] blocks" $OUT/bench_packed.log | tail -1 | cut -c1-700
JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B --storage packed2bit --steps 10 --groups 0 > $OUT/bench_packed_g0.json 2> $OUT/bench_packed_g0.log
grep "jwas_hip\] blocks" $OUT/bench_packed_g0.log | tail -1 | cut -c1-700
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d launch_us=%.2f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["avg_launch_us"]))
PY
done
