#!/bin/bash
# Development probe (GPU box): a parity subset, then the sampler's phase counters (JWAS_HIP_DEBUG_PHASES) for the workloads
# given as arguments, e.g.  bash scripts/dev_probe.sh "--workload refbench" "--pi-fixed 0.95"
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -3
B="--no-cpu-baseline --via-api 0 --steps 10 --warmup 10 --burnin 0"
for w in "$@"; do
JWAS_HIP_DEBUG_PHASES=1 timeout 300 python bench.py $B $w 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['name'], d['value'], d['ms_per_step'], d['config']['device_sweep_ms'])"; grep jwas_hip /tmp/err.txt | tail -1
done
