#!/bin/bash
# Builds libjbonsai_amd.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: MLPG/LF0 must keep the reference's rounding; the vocoder
# writes its fused multiply-adds explicitly.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libjbonsai_amd.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value"
mkdir -p build
objs=()
for f in jb_mlpg.hip jb_gv_gang.hip jb_vocoder.hip jb_mglsa.hip jb_postfilter.hip jb_batch.cpp jb_voice.cpp jb_engine.cpp jb_multi.cpp; do
  [ -f "$f" ] || continue
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ jb_device.h -nt "$o" ] || [ jb_host.h -nt "$o" ] \
     || [ ../../include/jbonsai_amd.h -nt "$o" ] || { [ -f jb_voice.h ] && [ jb_voice.h -nt "$o" ]; }; then
    echo "hipcc $f"
    $HIPCC $FLAGS -x hip -c "$f" -o "$o"
  fi
  objs+=("$o")
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o $OUT "${objs[@]}"
echo "built $(realpath $OUT)"
