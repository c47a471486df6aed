#!/bin/bash
# Collect the round's measurement artefacts on the GPU box:  bash scripts/profile_round.sh r01
# (plain bench line, rocprofv3 kernel trace + stats of the same command, FETCH_SIZE / WRITE_SIZE in separate --pmc passes).
# Outputs land in gpurun_out/<tag>/; scripts/summarize_profiles.py condenses them into the files kept under profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python -m pytest tests -m gpu -q 2>&1 | tail -5 > "$OUT/gpu_tests.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -o bench -- python bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/ktrace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python bench.py --no-cpu-baseline --steps 3 > "$OUT/bench_under_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python bench.py --no-cpu-baseline --steps 3 > "$OUT/bench_under_pmc_write.json" 2> "$OUT/pmc_write.err"
python scripts/summarize_profiles.py "$OUT" 30 3
# the raw traces are large: keep only the summaries
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
ls -la "$OUT"
