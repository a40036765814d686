#!/bin/bash
# Builds libsceneego_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libsceneego_hip.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $*"
mkdir -p _obj
pids=()
newer() {  # source $1 or a shared header newer than object $2
  [ ! -f $2 ] || [ $1 -nt $2 ] || [ common.h -nt $2 ] || [ conv_common.h -nt $2 ] || [ bf16_common.h -nt $2 ] || [ ../../include/sceneego_hip.h -nt $2 ]
}
for f in voxelize gather softargmax conv3d conv3d_tiled conv3d_wino conv3d_bf16 conv3d_bf16_tiled; do
  extra=""
  [ "$f" = voxelize ] && extra="-ffp-contract=off"
  if newer $f.hip _obj/$f.o; then
    hipcc $FLAGS $extra -c $f.hip -o _obj/$f.o &
    pids+=($!)
  fi
done
# the F(4,7) 7^3 kernel: one object per input layout (each takes minutes to compile: 390 unrolled MFMAs under sched_group_barrier)
for v in 0 1; do
  if newer conv3d_wino47.hip _obj/conv3d_wino47_$v.o || [ wino47_matrices.h -nt _obj/conv3d_wino47_$v.o ]; then
    hipcc $FLAGS -DSE_K7F_PLANAR=$v -c conv3d_wino47.hip -o _obj/conv3d_wino47_$v.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/*.o
echo "built $(realpath $OUT)"
