#!/bin/bash
# where the resident GV kernel spends its time: a -DJB_GG_PROFILE build of the library, one-stream run.
# The variant is built BEFOREHAND, where hipcc runs without a GPU (tools/build_variant.sh ggprof -DJB_GG_PROFILE=1
# [more -D flags] -> tools/_ab_ggprof/libjbonsai_amd.so, which travels to the GPU box); this script only swaps it
# in for one bench run and puts the product library back.
cd "$(dirname "$0")/.."
V=${1:-tools/_ab_ggprof/libjbonsai_amd.so}
[ -f "$V" ] || { echo "$V missing: run tools/build_variant.sh ggprof -DJB_GG_PROFILE=1 first"; exit 1; }
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep_ggprof.so
trap 'cp /tmp/_keep_ggprof.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp "$V" jbonsai_amd/libjbonsai_amd.so
JB_ONE_STREAM=1 JB_GG_PROFILE_PRINT=1 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>&1 | grep -A 300 "ticks" | head -330
