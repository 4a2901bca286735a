#!/bin/bash
# Builds libjbonsai_amd.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: MLPG/LF0 must keep the reference's rounding; the vocoder
# writes its fused multiply-adds explicitly.
# Stale objects are compiled side by side (a from-scratch build: 80 s one after the other, ~25 s this way).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libjbonsai_amd.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value"
mkdir -p build
objs=()
pids=()
names=()
for f in jb_mlpg.hip jb_gv_gang.hip jb_vocoder.hip jb_mglsa.hip jb_postfilter.hip jb_batch.cpp jb_voice.cpp jb_engine.cpp jb_multi.cpp; do
  [ -f "$f" ] || continue
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ jb_device.h -nt "$o" ] || [ jb_host.h -nt "$o" ] \
     || [ ../../include/jbonsai_amd.h -nt "$o" ] || { [ -f jb_voice.h ] && [ jb_voice.h -nt "$o" ]; }; then
    echo "hipcc $f"
    # (into a temporary name: an interrupted or failed compile must not leave a fresh-looking object behind)
    ( $HIPCC $FLAGS -x hip -c "$f" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
    pids+=($!)
    names+=("$f")
  fi
  objs+=("$o")
done
fail=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then
    echo "hipcc failed: ${names[$i]}" >&2
    fail=1
  fi
done
[ $fail -eq 0 ] || exit 1
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o $OUT "${objs[@]}"
echo "built $(realpath $OUT)"
