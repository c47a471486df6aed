cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mt or multitrait or mega" 2>&1 | tail -3
B="--no-cpu-baseline --via-api 0 --steps 10 --warmup 5 --burnin 0"
JWAS_HIP_DEBUG_PHASES=1 python bench.py $B --workload config4 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['device_sweep_ms'])"; grep jwas_hip /tmp/err.txt | tail -1
