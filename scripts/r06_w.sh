#!/bin/bash
# Same-box A/B after the PP template split (k_group_step<., ., PP>): prev = commit aa48826, new = split kernels.  Steady-state paths
# plus the two sampler-bound workloads the late hand-over / cooperative apply were built for, and the knobs one by one on config 3 steady.
cd $GRAFT_REPO_ROOT
L=jwas.jl_amd/csrc/libjwas_hip.so
cp $L /tmp/new.so
B="--no-cpu-baseline --via-api 0"
run() {  # label, env, args
  env $2 python bench.py $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(12), '$3'.ljust(52), 'it/s=%.2f ms=%.2f launch_us=%.2f ev=%.0f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['events_per_sweep']))" 2>&1 | tail -1
}
S3="--workload config3 --burnin 1400 --warmup 0 --steps 30"
for v in new prev new prev; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp jwas.jl_amd/csrc/_dev/libjwas_hip_$v.so $L; fi
  run $v X=1 "$S3"
  run $v X=1 "--workload config2"
  run $v X=1 "--storage packed2bit"
done
cp /tmp/new.so $L
run new-coop0 JWAS_HIP_GROUP_COOP=0 "$S3"
run new-nolate JWAS_HIP_COMPACT_OFF=4 "$S3"
run new-both "JWAS_HIP_GROUP_COOP=0 JWAS_HIP_COMPACT_OFF=4" "$S3"
run new X=1 "--workload config3"
run new X=1 "--workload config3 --pi-fixed 0.95"
run new-coop0 JWAS_HIP_GROUP_COOP=0 "$S3"
run new-nolate JWAS_HIP_COMPACT_OFF=4 "$S3"
run new X=1 "$S3"
