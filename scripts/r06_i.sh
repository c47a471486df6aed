#!/bin/bash
# Round 6: ping-pong with one workgroup per block (2 or 4 blocks per launch) -- parity forced on / default; packed regression A/B.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_i; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
JWAS_HIP_PINGPONG=1 timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py -m gpu -q -n 4 -x 2>&1 | grep -v "$F" | tail -8 > $OUT/gpu_tests_pp1.log
tail -6 $OUT/gpu_tests_pp1.log
timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py tests/test_gpu_e2e.py -m gpu -q -n 4 -x 2>&1 | grep -v "$F" | tail -5 > $OUT/gpu_tests_default.log
tail -3 $OUT/gpu_tests_default.log
B="--no-cpu-baseline --via-api 0"
timeout 300 python bench.py $B --storage packed2bit --steps 30 > $OUT/bench_packed.json 2> $OUT/bench_packed.log
JWAS_BENCH_GROUPS_SMALL=4 timeout 300 python bench.py $B --workload config3 --steps 20 > $OUT/bench_config3_pp4.json 2> $OUT/bench_config3_pp4.log
JWAS_BENCH_GROUPS_SMALL=4 timeout 300 python bench.py $B --workload config2 --pi-fixed 0.95 --steps 20 > $OUT/bench_pifixed_pp4.json 2> $OUT/bench_pifixed_pp4.log
timeout 300 python bench.py $B --workload config3 --steps 20 > $OUT/bench_config3_pp2.json 2> $OUT/bench_config3_pp2.log
timeout 300 python bench.py $B --workload config2 --steps 30 > $OUT/bench_config2.json 2> $OUT/bench_config2.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d frac=%.3f launch_us=%.2f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
