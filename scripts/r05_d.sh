#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_d; mkdir -p $OUT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_rule_t.py -x -v -k "dense or literal or rule or pi0 or all_included or cooperative" > $OUT/tests_full.log 2>&1
grep -n "PASSED\|FAILED\|Fatal\|fault\|Memory access\|test_" $OUT/tests_full.log | tail -25 | cut -c1-220
