set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2a/gpu_tests.log
tail -5 gpurun_out/r2a/gpu_tests.log
export JWAS_BENCH_VERBOSE=1
timeout 600 python bench.py > gpurun_out/r2a/bench_config2.json 2> gpurun_out/r2a/bench_config2.log
timeout 600 python bench.py --pi-fixed 0.95 --no-cpu-baseline > gpurun_out/r2a/bench_config2_pifixed.json 2> gpurun_out/r2a/bench_config2_pifixed.log
timeout 600 python bench.py --workload config3 > gpurun_out/r2a/bench_config3.json 2> gpurun_out/r2a/bench_config3.log
timeout 600 python bench.py --workload config4 > gpurun_out/r2a/bench_config4.json 2> gpurun_out/r2a/bench_config4.log
timeout 600 python bench.py --workload config4 --mt-prior sparse --no-cpu-baseline > gpurun_out/r2a/bench_config4_sparse.json 2> gpurun_out/r2a/bench_config4_sparse.log
timeout 600 python bench.py --workload config5shard > gpurun_out/r2a/bench_config5shard.json 2> gpurun_out/r2a/bench_config5shard.log
timeout 600 python bench.py --workload refbench --no-cpu-baseline > gpurun_out/r2a/bench_refbench.json 2> gpurun_out/r2a/bench_refbench.log
for f in gpurun_out/r2a/bench_*.json; do echo $f; cat $f | head -c 1500; echo; done
nproc; lscpu | head -20
