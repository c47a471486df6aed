#!/bin/bash
# Same-box A/B: the library at the start of round 6 (csrc/_dev/libjwas_hip_r05.so, built from commit d628f15) against the current one.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_l; mkdir -p $OUT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
B="--no-cpu-baseline --via-api 0"
for v in new old new old; do
  if [ $v = old ]; then cp jwas.jl_amd/csrc/_dev/libjwas_hip_r05.so $L; else cp /tmp/new.so $L; fi
  for w in "--storage packed2bit" "--workload config2" "--workload refbench --warmup 10 --burnin 0" "--workload config5shard"; do
    JWAS_BENCH_GROUPS_SMALL=0 python bench.py $B --groups 4 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$w'.ljust(44), 'it/s=%.2f ms=%.2f launch_us=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))" 2>&1 | tail -1
  done
done
cp /tmp/new.so $L
