#!/bin/bash
# Same-box A/B of PREBUILT libraries (built locally from different sources, e.g. under tools/_ab_base/, which
# travels to the GPU box but is git-ignored): tools/ab_libs.sh libA.so libB.so ...  -- step, parameter-generation
# part and vocoder kernel of bench.py, twice each, alternating.  No rocprofv3 involved.
cd "$(dirname "$0")/.."
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep.so
trap 'cp /tmp/_keep.so jbonsai_amd/libjbonsai_amd.so' EXIT
for rep in $(seq 1 ${REPS:-2}); do for l in "$@"; do
  cp $l jbonsai_amd/libjbonsai_amd.so
  timeout -k 5 120 python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', 'step', round(d['ms_per_step'], 2), 'pg', round(d['ms_per_step'] - d['roofline']['kernel_ms'], 2), 'voc', round(d['roofline']['kernel_ms'], 2))"
done; done
