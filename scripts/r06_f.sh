#!/bin/bash
# Round 6, sixth GPU call: the DEFERRED APPLY of the grouped launches (jwas_hip_setup_groups_ex) -- parity, then the benches.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_f; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 1200 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py -m gpu -q -n 4 -x 2>&1 | grep -v "$F" | tail -12 > $OUT/gpu_tests.log
tail -12 $OUT/gpu_tests.log
JWAS_HIP_PINGPONG=1 timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_packed.py -m gpu -q -n 4 -x 2>&1 | grep -v "$F" | tail -4 > $OUT/gpu_tests_pp1.log
tail -4 $OUT/gpu_tests_pp1.log
B="--no-cpu-baseline --via-api 0"
export JWAS_BENCH_DEFERRED=1
JWAS_BENCH_GROUPS_SMALL=2 timeout 300 python bench.py $B --workload config3 --steps 20 > $OUT/bench_config3_pp_def.json 2> $OUT/bench_config3_pp_def.log
JWAS_BENCH_GROUPS_SMALL=2 timeout 300 python bench.py $B --workload config2 --pi-fixed 0.95 --steps 20 > $OUT/bench_pifixed_pp_def.json 2> $OUT/bench_pifixed_pp_def.log
timeout 300 python bench.py $B --workload config2 --steps 30 > $OUT/bench_config2_def.json 2> $OUT/bench_config2_def.log
timeout 300 python bench.py $B --workload config2 --steps 30 --groups 2 > $OUT/bench_config2_def_g2.json 2> $OUT/bench_config2_def_g2.log
unset JWAS_BENCH_DEFERRED
timeout 300 python bench.py $B --workload config2 --steps 30 > $OUT/bench_config2.json 2> $OUT/bench_config2.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d m=%d frac=%.3f gsetup=%.1f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], c["blocks_per_launch"], d["roofline"]["frac"], c.get("group_setup_s",0)))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
