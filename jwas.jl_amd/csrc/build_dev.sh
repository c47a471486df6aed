#!/bin/bash
# Development build (role switches, update-role phase clocks: -DJWAS_HIP_DEV_KNOBS) into _dev/libjwas_hip.so; never shipped.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DJWAS_HIP_DEV_KNOBS $JWAS_DEV_EXTRA"
mkdir -p _dev
"$HIPCC" $FLAGS "$@" -c jwas_hip.hip -o _dev/jwas_hip.o &
P1=$!
"$HIPCC" $FLAGS "$@" -c resident_launch.hip -o _dev/resident_launch.o &
P2=$!
wait $P1; wait $P2
"$HIPCC" --offload-arch=gfx950 -fPIC -shared _dev/jwas_hip.o _dev/resident_launch.o -o _dev/libjwas_hip.so
rm -f _dev/*.o
