#!/bin/bash
# Same-box A/B: the library before the late hand-over / cooperative apply (csrc/_dev/libjwas_hip_prev.so, commit aa48826) against the
# final one, on the STEADY-STATE paths those changes should not touch.
cd $GRAFT_REPO_ROOT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
B="--no-cpu-baseline --via-api 0"
for v in new prev new prev; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  for w in "--workload config3 --burnin 1400 --warmup 0 --steps 30" "--workload config2" "--storage packed2bit"; do
    python bench.py $B $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.ljust(5), '$w'.ljust(52), 'it/s=%.2f ms=%.2f launch_us=%.2f ev=%.0f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['events_per_sweep']))" 2>&1 | tail -1
  done
done
cp /tmp/new.so $L
