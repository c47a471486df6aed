# quick GPU check: parity tests + short benches of the sampler-bound workloads (no CPU baseline)
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-quick}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "tests rc=$?" >> $OUT/gpu_tests.log
tail -15 $OUT/gpu_tests.log
export JWAS_BENCH_VERBOSE=1
B="--no-cpu-baseline --via-api 0 --steps 10"
timeout 300 python bench.py $B > $OUT/bench_config2.json 2> $OUT/bench_config2.log
timeout 300 python bench.py $B --pi-fixed 0.95 --warmup 10 --burnin 0 > $OUT/bench_config2_pifixed.json 2> $OUT/bench_config2_pifixed.log
timeout 300 python bench.py $B --workload config3 --warmup 10 --burnin 0 > $OUT/bench_config3.json 2> $OUT/bench_config3.log
timeout 300 python bench.py $B --workload config4 --warmup 5 --burnin 0 > $OUT/bench_config4.json 2> $OUT/bench_config4.log
timeout 300 python bench.py $B --workload config4 --mt-prior sparse --warmup 10 --burnin 0 > $OUT/bench_config4_sparse.json 2> $OUT/bench_config4_sparse.log
timeout 300 python bench.py $B --workload refbench --warmup 5 --burnin 0 > $OUT/bench_refbench.json 2> $OUT/bench_refbench.log
JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B --pi-fixed 0.95 --warmup 10 --burnin 0 --steps 3 2>&1 | grep jwas_hip | tail -3
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f in_model=%.0f bs=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["markers_in_model"], c["block_size"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
