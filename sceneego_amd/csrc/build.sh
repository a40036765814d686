#!/bin/bash
# Builds libsceneego_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libsceneego_hip.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $*"
mkdir -p _obj
pids=()
for f in voxelize gather softargmax conv3d conv3d_tiled conv3d_wino conv3d_bf16 conv3d_bf16_tiled; do
  extra=""
  [ "$f" = voxelize ] && extra="-ffp-contract=off"
  if [ ! -f _obj/$f.o ] || [ $f.hip -nt _obj/$f.o ] || [ common.h -nt _obj/$f.o ] || [ conv_common.h -nt _obj/$f.o ] || [ bf16_common.h -nt _obj/$f.o ] || [ ../../include/sceneego_hip.h -nt _obj/$f.o ]; then
    hipcc $FLAGS $extra -c $f.hip -o _obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/*.o
echo "built $(realpath $OUT)"
