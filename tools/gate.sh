#!/bin/bash
# Ceiling of what removing kernels from the step can be worth (GPU box): the library built with
# -DJB_DBG_GATES (tools/build_variant.sh gates -DJB_DBG_GATES) leaves out the launches of JB_DBG_SKIP's bit
# mask after bench.py's warm-up steps (jb_device.h); their outputs stay valid from the warm-up.
#   tools/gate.sh [masks...]      default: 0 1 2 3 4 8 12 16
cd "$(dirname "$0")/.."
lib=tools/_ab_gates/libjbonsai_amd.so
[ -f $lib ] || { echo "build it first: tools/build_variant.sh gates -DJB_DBG_GATES"; exit 2; }
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep.so
trap 'cp /tmp/_keep.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp $lib jbonsai_amd/libjbonsai_amd.so
masks=${@:-0 1 2 3 4 8 12 16 0}
for m in $masks; do
  JB_DBG_SKIP=$m JB_DBG_SKIP_AFTER=${AFTER:-3} timeout -k 5 120 python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skip mask $m: step', round(d['ms_per_step'], 2), 'pg', round(d['ms_per_step'] - d['roofline']['kernel_ms'], 2), 'voc', round(d['roofline']['kernel_ms'], 2), 'redone', d['config']['chunks_redone_last_step'])"
done
