#!/bin/bash
# First checkpoint of the redo round at 32 / 40 / 48 / 56 frames (JB_DBG_CKPT1 of the -DJB_DBG_GATES library:
# tools/build_variant.sh gates -DJB_DBG_GATES) on a ragged batch and on 1024 distinct utterances, with the trace of
# the rounds (JB_REDO_TRACE=1): which stages a round went through and what the step took.
cd "$(dirname "$0")/.."
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep.so
trap 'cp /tmp/_keep.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp tools/_ab_gates/libjbonsai_amd.so jbonsai_amd/libjbonsai_amd.so
for c in 32 40 48 56; do
  echo "== first checkpoint $c"
  JB_DBG_CKPT1=$c JB_REDO_TRACE=1 python bench.py --mixed --batch 512 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | grep -a "second checkpoint\|ms_per_step" | sed 's/.*"ms_per_step": \([0-9.]*\).*/step \1/' | tail -3
  JB_DBG_CKPT1=$c JB_REDO_TRACE=1 python bench.py --batch 1024 --frames 6386 --distinct 1024 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | grep -a "second checkpoint\|ms_per_step" | sed 's/.*"ms_per_step": \([0-9.]*\).*/step \1/' | tail -3
done
