#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing up\|amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "skip_and_verify" 2>&1 | grep -v "$F" | tail -25
