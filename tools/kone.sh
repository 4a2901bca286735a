#!/bin/bash
# one-stream (alone) kernel times of prebuilt libraries, same box: tools/kone.sh libA.so libB.so ...   (PAT = kernel grep pattern)
cd "$(dirname "$0")/.."
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep_kone.so
trap 'cp /tmp/_keep_kone.so jbonsai_amd/libjbonsai_amd.so' EXIT
for l in "$@"; do
  cp "$l" jbonsai_amd/libjbonsai_amd.so
  echo "== $l"
  JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh 2>&1 | grep -E "${PAT:-gv_gang}|rror"
done
