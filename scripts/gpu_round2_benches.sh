# every bench workload with its CPU baseline (N = 1)
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2bench}
mkdir -p $OUT
export JWAS_BENCH_VERBOSE=1
timeout 200 python bench.py --cpu-seconds 5 > $OUT/bench_config2.json 2> $OUT/bench_config2.log
timeout 200 python bench.py --pi-fixed 0.95 --cpu-seconds 3 > $OUT/bench_config2_pifixed.json 2> $OUT/bench_config2_pifixed.log
timeout 200 python bench.py --workload config3 --cpu-seconds 4 > $OUT/bench_config3.json 2> $OUT/bench_config3.log
timeout 200 python bench.py --workload config4 --cpu-seconds 4 > $OUT/bench_config4.json 2> $OUT/bench_config4.log
timeout 200 python bench.py --workload config4 --mt-prior sparse --no-cpu-baseline > $OUT/bench_config4_sparse.json 2> $OUT/bench_config4_sparse.log
timeout 200 python bench.py --workload config5shard --cpu-seconds 4 > $OUT/bench_config5shard.json 2> $OUT/bench_config5shard.log
timeout 200 python bench.py --workload refbench --cpu-seconds 4 > $OUT/bench_refbench.json 2> $OUT/bench_refbench.log
timeout 200 python bench.py --storage packed2bit --no-cpu-baseline > $OUT/bench_config2_packed.json 2> $OUT/bench_config2_packed.log
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    cb=d.get("cpu_baseline",{}); va=d.get("via_api",{})
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f in_model=%.0f bs=%d frac=%.3f cpu=%s via_api=%s" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["markers_in_model"], c["block_size"], d["roofline"]["frac"], cb.get("value"), va.get("value")))
    if cb: print("   ", cb["cores"], cb["sample"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
