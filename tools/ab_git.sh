#!/bin/bash
# A/B of the working tree's jb_vocoder.hip against a saved copy (tools/_ab_head_voc.hip.txt, made
# locally with `git show HEAD:...`; the GPU box has no .git) on one box: bench twice each.
cd "$(dirname "$0")/.."
cp jbonsai_amd/csrc/jb_vocoder.hip /tmp/voc_new.hip
for v in head new head new; do
  if [ $v = head ]; then cp tools/_ab_head_voc.hip.txt jbonsai_amd/csrc/jb_vocoder.hip; else cp /tmp/voc_new.hip jbonsai_amd/csrc/jb_vocoder.hip; fi
  (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && ./build.sh >/dev/null 2>&1)
  echo "== $v"
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['chunks_redone_last_step'])"
done
cp /tmp/voc_new.hip jbonsai_amd/csrc/jb_vocoder.hip; (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && ./build.sh >/dev/null 2>&1)
