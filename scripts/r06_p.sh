#!/bin/bash
# Round 6: the multi-trait sparse steady state on 1024-marker blocks (draws parked in LDS up to 1024 x 3)?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_p; mkdir -p $OUT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -n 4 -k "mt or multitrait or mega or random" 2>&1 | grep -v "$F" | tail -4 > $OUT/gpu_tests_mt.log; tail -3 $OUT/gpu_tests_mt.log
B="--no-cpu-baseline --via-api 0 --workload config4 --mt-prior sparse --steps 20"
for bs in 512 1024; do
  JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B --block-size $bs > $OUT/bench_sparse_$bs.json 2> $OUT/bench_sparse_$bs.log
  grep "jwas_hip\] blocks" $OUT/bench_sparse_$bs.log | tail -1 | cut -c1-330
  python - $OUT/bench_sparse_$bs.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d launch_us=%.2f frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
