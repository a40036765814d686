#!/bin/bash
# Builds libsceneego_hip.so for gfx950 (cross-compiles without a GPU).
# Usage: build.sh [--devtools] [extra hipcc flags]
#   --devtools  builds ../libsceneego_hip_dev.so instead (tools/ load it through SCENEEGO_HIP_LIB) with -DSE_DEVTOOLS: the A/B kernel selector (se_debug_set_variant), the retired kernel variants it selects
#               (devtools/*.inc, included only under SE_DEVTOOLS) and the cycle-stamp hooks used by tools/.  The production library is built WITHOUT it.
# Objects live in _obj/<key>/ where <key> hashes the compiler version and the flag line, so objects of another compiler or
# another flag set are never reused; inside a key a source is rebuilt when it or a shared header is newer than its object.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libsceneego_hip.so
DEV=""
if [ "${1:-}" = "--devtools" ]; then DEV="-DSE_DEVTOOLS"; OUT=../libsceneego_hip_dev.so; shift; fi
FLAGS="-O3 $DEV --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $*"
KEY=$( (hipcc --version 2>/dev/null; echo "$FLAGS") | sha256sum | cut -c1-12)
OBJ=_obj/$KEY
mkdir -p "$OBJ"
pids=()
newer() {  # source $1, a shared header or (development builds) a devtools/ include newer than object $2
  [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ common.h -nt "$2" ] || [ conv_common.h -nt "$2" ] || [ bf16_common.h -nt "$2" ] || [ wino67_matrices.h -nt "$2" ] || [ fft24.h -nt "$2" ] || [ ../../include/sceneego_hip.h -nt "$2" ] && return 0
  if [ -n "$DEV" ]; then for i in devtools/*.inc; do [ "$i" -nt "$2" ] && return 0; done; fi
  return 1
}
OBJS=()
for f in voxelize gather softargmax conv2d_1x1 conv2d_3x3 conv3d conv3d_tiled conv3d_wino conv3d_wino2d conv3d_wino44pp conv3d_wino67 conv3d_fft7 conv3d_bf16 conv3d_bf16_tiled conv3d_split; do
  [ -f $f.hip ] || { echo "build.sh: source $f.hip is missing" >&2; exit 1; }
  OBJS+=($OBJ/$f.o)
  extra=""
  [ "$f" = voxelize ] && extra="-ffp-contract=off"
  # no SLP packing of float32 arithmetic into v_pk_*_f32: packed VALU beside an MFMA stream is an anti-lever (see commit() there)
  [ "$f" = conv3d_wino2d ] && extra="-fno-slp-vectorize"
  [ "$f" = conv3d_wino44pp ] && extra="-fno-slp-vectorize -Wno-inline-asm"      # -Wno-inline-asm: the "m0" clobber of the LDS-DMA asm (reserved register)
  [ "$f" = conv3d_wino67 ] && extra="-Wno-inline-asm"
  if newer $f.hip $OBJ/$f.o; then
    hipcc $FLAGS $extra -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
# development builds: the F(4,3) x F(4,3) experiment of round 3 (se_debug_set_variant(63))
if [ -n "$DEV" ]; then
  OBJS+=($OBJ/conv3d_wino44.o)
  if newer conv3d_wino44.hip $OBJ/conv3d_wino44.o; then
    hipcc $FLAGS -c conv3d_wino44.hip -o $OBJ/conv3d_wino44.o &
    pids+=($!)
  fi
fi
# the F(4,7) 7^3 kernel: one object per input layout (each takes minutes to compile: 390 unrolled MFMAs under sched_group_barrier)
for v in 0 1; do
  OBJS+=($OBJ/conv3d_wino47_$v.o)
  if newer conv3d_wino47.hip $OBJ/conv3d_wino47_$v.o || [ wino47_matrices.h -nt $OBJ/conv3d_wino47_$v.o ]; then
    hipcc $FLAGS -DSE_K7F_PLANAR=$v -c conv3d_wino47.hip -o $OBJ/conv3d_wino47_$v.o &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait $p || rc=1; }; done
[ $rc -eq 0 ] || { echo "build.sh: compilation failed" >&2; exit 1; }
# the kernels with hand-counted vmcnt waits around inline-assembly LDS-DMAs: no VGPR spill, no scratch, M0 written only
# by the asm - or the build fails (check_codeobj.py says why).  Attribution builds (extra flags) may spill: checked, not fatal, there.
for f in conv3d_wino44pp conv3d_wino67; do
  crc=0; python3 check_codeobj.py $OBJ/$f.o || crc=$?
  if [ $crc -eq 2 ]; then echo "build.sh: the code-object check of $f.hip could not run (tools / metadata, see above; SE_SKIP_CODEOBJ_CHECK=1 skips it)" >&2; exit 1; fi
  if [ $crc -ne 0 ]; then
    if [ -z "$*" ]; then echo "build.sh: $f.hip violates the conditions its hand-counted waits rely on" >&2; exit 1; fi
    echo "build.sh: (extra flags given: continuing)" >&2
  fi
done
# explicit object list: a stale object of a removed / renamed source in the same key directory is never linked
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT "${OBJS[@]}"
# the hash of the sources this binary was built from (sceneego_amd/_lib.py: built_fingerprint / source_fingerprint): counter records
# under profiles/ carry it, bench.py prints their traffic figure only for the binary they were taken on
python3 - "$OUT" <<'PY'
import hashlib, os, sys
here = os.getcwd()
h = hashlib.sha256()
files = sorted(os.path.join(here, f) for f in os.listdir(here) if f.endswith((".hip", ".h", ".sh")))
files.append(os.path.join(here, "..", "..", "include", "sceneego_hip.h"))
for p in files:
    h.update(os.path.basename(p).encode() + b"\0")
    h.update(open(p, "rb").read())
open(os.path.splitext(sys.argv[1])[0] + ".srchash", "w").write(h.hexdigest()[:16] + "\n")
PY
echo "built $(realpath $OUT) (objects: $OBJ)"
