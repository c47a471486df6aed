#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -k "loopback or sharded or shard" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|bringing" | tail -15
