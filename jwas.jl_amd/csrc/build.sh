#!/bin/bash
# Builds libjwas_hip.so for gfx950 in-tree (the .so travels with the gpurun snapshot).
# -ffp-contract=off: the arithmetic contract shared with the oracle (no implicit FMA contraction).
# One translation unit per sampler family (step_launch.hpp) + jwas_hip.hip (context, C ABI, every other kernel), compiled in
# parallel; an object is rebuilt only when it is older than a source it includes (INCREMENTAL=0 forces everything).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJ=${JWAS_OBJ_DIR:-_obj}
OUT=${JWAS_OUT:-libjwas_hip.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $JWAS_EXTRA_FLAGS"
UNITS="jwas_hip step_st step_mtc1 step_mtb1 step_mt2 step_mega"
mkdir -p "$OBJ"
echo "$FLAGS $*" > "$OBJ/.flags.new"
if [ "${INCREMENTAL:-1}" = 0 ] || ! cmp -s "$OBJ/.flags.new" "$OBJ/.flags"; then rm -f "$OBJ"/*.o; fi
mv "$OBJ/.flags.new" "$OBJ/.flags"
needs() {      # does unit $1 have to be compiled?
    local o="$OBJ/$1.o"
    [ -f "$o" ] || return 0
    local deps="$1.hip ../../include/jwas_hip.h"
    case $1 in
        jwas_hip) deps="$deps $(ls *.hpp)" ;;
        step_st)  deps="$deps kernels.hpp rng.hpp sweep.hpp update_role.hpp sampler_common.hpp sampler_st.hpp sampler_mt.hpp step_launch.hpp step_launch_impl.hpp" ;;
        *)        deps="$deps kernels.hpp rng.hpp sweep.hpp update_role.hpp sampler_common.hpp sampler_st.hpp sampler_mt.hpp step_launch.hpp step_launch_impl.hpp" ;;
    esac
    for d in $deps; do [ "$d" -nt "$o" ] && return 0; done
    return 1
}
pids=""
for u in $UNITS; do
    if needs $u; then
        "$HIPCC" $FLAGS "$@" -c $u.hip -o "$OBJ/$u.o" &
        pids="$pids $!"
    fi
done
for p in $pids; do wait $p; done
objs=""
for u in $UNITS; do objs="$objs $OBJ/$u.o"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared $objs -o "$OUT"
