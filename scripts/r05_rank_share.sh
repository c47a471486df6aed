#!/bin/bash
# What ONE rank of an 8-GPU config-2 job does per iteration with the round-5 default schedule (4 blocks per launch) and with one
# block per launch, plain and through jwas_hip_sweep_sharded on a one-rank RCCL communicator (cf. scripts/r04_rank_share.sh)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_rank_share; mkdir -p $out
B="--workload config2 --p 75000 --steps 200 --warmup 30 --burnin 60 --no-cpu-baseline --via-api 0"
for g in 4 2 0; do
  timeout 600 python bench.py $B --groups $g > $out/plain_g$g.json 2> $out/plain.err
  timeout 600 python bench.py $B --groups $g --one-rank-comm > $out/sharded_g$g.json 2> $out/sharded.err
done
python - <<PY
import json
res = {}
for g in (4, 2, 0):
    for k in ("plain", "sharded"):
        d = json.loads([l for l in open("$out/%s_g%d.json" % (k, g)).read().splitlines() if l.startswith('{"metric"')][-1]); c = d["config"]
        res["%s_groups%d" % (k, g)] = {"ms_per_step": d["ms_per_step"], "device_sweep_ms": c["device_sweep_ms"], "host_ms_per_step": c["host_ms_per_step"],
                                       "sharded_path": c["sharded_path"], "events_per_sweep": c["events_per_sweep"], "blocks_per_launch": c["blocks_per_launch"],
                                       "roofline_frac": d["roofline"]["frac"], "avg_launch_us": d["roofline"]["avg_launch_us"]}
        print(k, g, res["%s_groups%d" % (k, g)])
json.dump(res, open("$out/rank_share.json", "w"), indent=1)
PY
