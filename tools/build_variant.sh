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
case $src in conv3d_wino2d) extra="-fno-slp-vectorize";; conv3d_wino44pp) extra="-fno-slp-vectorize -Wno-inline-asm";; conv3d_wino67) extra="-Wno-inline-asm";;
  voxelize) extra="-ffp-contract=off";; esac
mkdir -p _obj/variants
# never link a stale object: remove it first, stop when the compile fails (`| grep error || true` hid a failed compile and
# the old object of the same name was linked and reported as "built")
rm -f _obj/variants/${src}_$name.o
log=_obj/variants/${src}_$name.log
if ! hipcc $FLAGS $extra "$@" -c $src.hip -o _obj/variants/${src}_$name.o > "$log" 2>&1; then
  grep -E "error" "$log" >&2 || tail -20 "$log" >&2
  echo "build_variant.sh: compiling $src.hip ($name) FAILED - nothing linked" >&2
  exit 1
fi
case $src in conv3d_wino44pp|conv3d_wino67) python3 check_codeobj.py _obj/variants/${src}_$name.o || echo "build_variant.sh: (attribution / stamp variant: continuing)" >&2;; esac
objs=()
for o in $OBJ/*.o; do
  b=$(basename $o .o)
  if [ "$b" = "$src" ]; then objs+=(_obj/variants/${src}_$name.o); else objs+=($o); fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libse_$name.so "${objs[@]}"
echo "built sceneego_amd/libse_$name.so"
