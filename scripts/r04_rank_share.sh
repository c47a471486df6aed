#!/bin/bash
# what ONE rank of an 8-GPU config-2 job does per iteration, measured on one GPU (no curve can be measured
# on the one-GPU boxes): 75 000 of the 600 000 markers, (a) plain sweep, (b) through jwas_hip_sweep_sharded on a one-rank RCCL
# communicator (snapshot D2D, pack kernel, ncclAllReduce, apply kernel), + the kernel trace of (b).  -> gpurun_out/r04_rank_share/
out=gpurun_out/r04_rank_share; mkdir -p $out
B="--workload config2 --p 75000 --steps 200 --warmup 30 --burnin 60 --no-cpu-baseline --via-api 0"
timeout 600 python bench.py $B > $out/plain.json 2> $out/plain.err
timeout 600 python bench.py $B --one-rank-comm > $out/sharded.json 2> $out/sharded.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py $B --one-rank-comm > $GRAFT_REPO_ROOT/$out/sharded_prof.json 2> $GRAFT_REPO_ROOT/$out/sharded_prof.err
cd $GRAFT_REPO_ROOT
find $out/prof -type f | head -30
for f in $(find $out/prof -name "*stats*.csv"); do cp $f $out/$(basename $f); done
rm -rf $out/prof
python - <<PY
import json
for k in ("plain","sharded","sharded_prof"):
    try:
        d=json.loads([l for l in open("$out/%s.json"%k).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
        print(k, "ms/step", round(d["ms_per_step"],4), "device sweep", round(c["device_sweep_ms"],4), "host", round(c["host_ms_per_step"],4), "sharded", c["sharded_path"], "events", c["events_per_sweep"])
    except Exception as e: print(k, "FAILED", e)
PY
for f in $out/*stats*.csv; do echo "== $f"; head -14 $f | cut -c1-200; done
