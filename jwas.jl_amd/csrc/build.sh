#!/bin/bash
# Builds libjwas_hip.so for gfx950 in-tree (the .so travels with the gpurun snapshot).
# -ffp-contract=off: the arithmetic contract shared with the oracle (no implicit FMA contraction).
# Two translation units, compiled in parallel: jwas_hip.hip (context, C ABI, launch-per-block sweep) and
# resident_launch.hip (the resident-sampler sweep's kernels).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
mkdir -p _obj
"$HIPCC" $FLAGS "$@" -c jwas_hip.hip -o _obj/jwas_hip.o &
P1=$!
"$HIPCC" $FLAGS "$@" -c resident_launch.hip -o _obj/resident_launch.o &
P2=$!
wait $P1
wait $P2
"$HIPCC" --offload-arch=gfx950 -fPIC -shared _obj/jwas_hip.o _obj/resident_launch.o -o libjwas_hip.so
