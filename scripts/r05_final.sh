#!/bin/bash
# Round 5, final code: the whole GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_final; mkdir -p $OUT
python -m pytest tests -m gpu -q -n 4 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up" | tail -6 > $OUT/gpu_tests_final.log; cat $OUT/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $OUT/smoke.log; cat $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-400
