#!/bin/bash
# kernel trace of config 4 under Rule T (which kernels, how long)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="--no-cpu-baseline --via-api 0 --steps 10 --workload config4 --warmup 5 --burnin 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o bench -- python $GRAFT_REPO_ROOT/bench.py $B > $OUT/bench_under_rocprof.json 2> $OUT/ktrace.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/ktrace -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-220
python scripts/summarize_profiles.py $OUT 10 3 config4 2>&1 | tail -3
find $OUT -name "*kernel_trace.csv" -size +5M -delete; find $OUT -name "*.db" -delete
