#!/bin/bash
# Builds a VARIANT of the library beside the product: tools/build_variant.sh NAME [extra hipcc flags...]
#   -> tools/_ab_NAME/libjbonsai_amd.so (git-ignored, travels to the GPU box); objects under tools/_ab_NAME/obj.
# Used for same-box A/B runs of prebuilt libraries (tools/ab_libs.sh, tools/gate.sh).
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/_ab_$name; mkdir -p $out/obj
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value $*"
pids=()
for f in jb_mlpg.hip jb_gv_gang.hip jb_vocoder.hip jb_mglsa.hip jb_postfilter.hip jb_batch.cpp jb_voice.cpp jb_engine.cpp jb_multi.cpp; do
  $HIPCC $FLAGS -x hip -c jbonsai_amd/csrc/$f -o $out/obj/${f%.*}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o $out/libjbonsai_amd.so $out/obj/*.o
echo "built $out/libjbonsai_amd.so"
