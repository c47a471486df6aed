#!/bin/bash
# Round 5, first GPU call: the whole GPU suite on the refactored library, phase counters of the sampler-bound workloads (baseline
# of this round), the BayesR shard check at full size and one rank's share of config 3.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_a; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -15 > $OUT/gpu_tests.log
tail -5 $OUT/gpu_tests.log
B="--no-cpu-baseline --via-api 0 --steps 10"
export JWAS_HIP_DEBUG_PHASES=1
timeout 300 python bench.py $B --workload config4 --warmup 5 --burnin 0 > $OUT/bench_config4.json 2> $OUT/bench_config4.log
timeout 300 python bench.py $B --workload config4 --mt-method BayesB --warmup 5 --burnin 0 > $OUT/bench_config4_bayesb.json 2> $OUT/bench_config4_bayesb.log
timeout 300 python bench.py $B --workload refbench --warmup 5 --burnin 0 > $OUT/bench_refbench.json 2> $OUT/bench_refbench.log
timeout 300 python bench.py $B --storage packed2bit > $OUT/bench_packed.json 2> $OUT/bench_packed.log
unset JWAS_HIP_DEBUG_PHASES
timeout 300 python bench.py $B --steps 30 > $OUT/bench_config2.json 2> $OUT/bench_config2.log
for w in config4 config4_bayesb refbench packed; do echo "== $w"; grep "jwas_hip\] blocks" $OUT/bench_$w.log | tail -1 | cut -c1-600; done
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "it/s=%.2f ms=%.2f sweep_ms=%.2f events=%.0f bs=%d frac=%.3f" % (d["value"], d["ms_per_step"], c["device_sweep_ms"], c["events_per_sweep"], c["block_size"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 900 python scripts/shard_check.py --method BayesR --iters 100 --json $OUT/shard_check_bayesr.json > $OUT/shard_check_bayesr.log 2>&1
grep "second-half\|it100\|it20:" $OUT/shard_check_bayesr.log
S="--workload config3 --p 75000 --steps 100 --warmup 30 --burnin 40 --no-cpu-baseline --via-api 0"
timeout 600 python bench.py $S > $OUT/rank_share_config3_plain.json 2> $OUT/rank_share_config3_plain.err
timeout 600 python bench.py $S --one-rank-comm > $OUT/rank_share_config3_sharded.json 2> $OUT/rank_share_config3_sharded.err
for k in plain sharded; do python - $OUT/rank_share_config3_$k.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1]); c=d["config"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(d["ms_per_step"],4), "device sweep", round(c["device_sweep_ms"],4), "host", round(c.get("host_ms_per_step",0),4), "events", c["events_per_sweep"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
