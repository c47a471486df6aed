#!/bin/bash
# the row-shard test that fails now and then inside parallel runs with the ping-pong samplers forced on: its message
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
for i in 1 2 3 4; do JWAS_HIP_PINGPONG=1 timeout 600 python -m pytest tests/test_gpu_groups.py tests/test_gpu_literal.py tests/test_gpu_packed.py tests/test_gpu_e2e.py -m gpu -q -n 4 2>&1 | grep -v "$F" | grep -B40 "JwasHipError\|Error" | tail -60; done
