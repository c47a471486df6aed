#!/bin/bash
# Bisect of the packed grouped regression: library variants under csrc/_dev (same box).
cd $GRAFT_REPO_ROOT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
B="--no-cpu-baseline --via-api 0"
for v in new r05 notg notgrl new r05 notg notgrl; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  JWAS_BENCH_GROUPS_SMALL=0 python bench.py $B --groups 4 --storage packed2bit 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.ljust(10), 'it/s=%.2f ms=%.2f launch_us=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))" 2>&1 | tail -1
done
cp /tmp/new.so $L
