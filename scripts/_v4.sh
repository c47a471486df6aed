cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multitrait_dense or mt_ or mega" 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -12
B="--no-cpu-baseline --via-api 0 --steps 10 --warmup 10 --burnin 0"
for w in "--workload config4" "--workload config4 --mt-prior sparse"; do
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B $w 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['name'], d['value'], d['ms_per_step'], d['config']['device_sweep_ms'])"; grep jwas_hip /tmp/err.txt | tail -1
done
