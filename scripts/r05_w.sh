#!/bin/bash
# Round 5: phase counters of the sampler workgroup in the regimes it bounds (config 4 sparse steady state, config 3, fixed pi, packed, grouped headline)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_w; mkdir -p $OUT
ph() { tag=$1; shift; JWAS_HIP_DEBUG_PHASES=1 timeout 600 python bench.py --no-cpu-baseline --via-api 0 --steps 3 "$@" 2>&1 >/dev/null | grep "jwas_hip\] blocks" | tail -2 > $OUT/$tag.txt; echo "== $tag"; cat $OUT/$tag.txt; }
ph config4_sparse --workload config4 --mt-prior sparse
ph config3 --workload config3
ph config2_packed --storage packed2bit
ph config2_groups4 --groups 4
ph config2_plain --groups 0
