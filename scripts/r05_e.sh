#!/bin/bash
# 2-bit packed storage with its own update geometry (update_role_wide): parity, then the config-2 packed bench with phase counters
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_packed.py -x -q 2>&1 | tail -25 > $OUT/tests.log; cat $OUT/tests.log | cut -c1-200
B="--no-cpu-baseline --via-api 0 --steps 10"
JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py $B --storage packed2bit > $OUT/bench_packed.json 2> $OUT/bench_packed.log
timeout 600 python bench.py $B > $OUT/bench_dense.json 2> $OUT/bench_dense.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(".json", ".log")).read()[-1500:])
PY
done
grep "jwas_hip\] blocks" $OUT/bench_packed.log | tail -1 | cut -c1-420
