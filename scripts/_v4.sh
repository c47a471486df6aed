cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or geometry or chain or many" 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -2
B="--no-cpu-baseline --via-api 0 --steps 10 --warmup 10 --burnin 0"
for w in "--workload refbench" "--workload config4" "--pi-fixed 0.95" "--workload config3"; do
python bench.py $B $w 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['name'], d['value'], d['ms_per_step'], d['config']['device_sweep_ms'])"
done
python bench.py --no-cpu-baseline --via-api 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['name'], d['value'], d['ms_per_step'], d['config']['device_sweep_ms'])"
