#!/bin/bash
# Builds libjwas_hip.so for gfx950 in-tree (the .so travels with the gpurun snapshot).
# -ffp-contract=off: the arithmetic contract shared with the oracle (no implicit FMA contraction).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" jwas_hip.hip -o libjwas_hip.so
