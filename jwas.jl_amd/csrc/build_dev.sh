#!/bin/bash
# Development build (role switches, update-role phase clocks: -DJWAS_HIP_DEV_KNOBS) into _dev/libjwas_hip.so; never shipped.
# JWAS_DEV_EXTRA adds flags (e.g. -DJWAS_HIP_COOP_RELACQ, -DJWAS_HIP_POISON_LDS=0x11111111).
set -e
cd "$(dirname "$0")"
mkdir -p _dev
JWAS_OBJ_DIR=_dev/obj JWAS_OUT=_dev/libjwas_hip.so JWAS_EXTRA_FLAGS="-DJWAS_HIP_DEV_KNOBS $JWAS_DEV_EXTRA" bash ./build.sh "$@"
