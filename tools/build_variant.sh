#!/bin/bash
# Development helper: a variant of ONE kernel source inside an otherwise standard development build, for A/B runs of several
# builds in one process (tools/ab_libs.py) or stamp / attribution builds.
#   tools/build_variant.sh NAME SOURCE [extra hipcc flags]   ->  sceneego_amd/libse_NAME.so
# e.g. tools/build_variant.sh stamp conv3d_wino44pp -DSE_STAMP44P
set -euo pipefail
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../sceneego_amd/csrc"
[ -f ../libsceneego_hip_dev.so ] || bash build.sh --devtools > /dev/null
FLAGS="-O3 -DSE_DEVTOOLS --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
KEY=$( (hipcc --version 2>/dev/null; echo "$FLAGS ") | sha256sum | cut -c1-12)
OBJ=_obj/$KEY
[ -d "$OBJ" ] || { echo "development objects $OBJ missing: run build.sh --devtools" >&2; exit 1; }
extra=""
case $src in conv3d_wino2d|conv3d_wino44pp) extra="-fno-slp-vectorize";; voxelize) extra="-ffp-contract=off";; esac
mkdir -p _obj/variants
hipcc $FLAGS $extra "$@" -c $src.hip -o _obj/variants/${src}_$name.o 2>&1 | grep -E "error" || true
objs=()
for o in $OBJ/*.o; do
  b=$(basename $o .o)
  if [ "$b" = "$src" ]; then objs+=(_obj/variants/${src}_$name.o); else objs+=($o); fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libse_$name.so "${objs[@]}"
echo "built sceneego_amd/libse_$name.so"
