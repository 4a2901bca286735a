#!/bin/bash
# Schedule experiments with the -DJB_DBG_GATES library (tools/gate.sh): JB_DBG_SCHED bit masks, twice each
cd "$(dirname "$0")/.."
lib=tools/_ab_gates/libjbonsai_amd.so
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep.so
trap 'cp /tmp/_keep.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp $lib jbonsai_amd/libjbonsai_amd.so
for rep in 1 2; do for m in ${@:-0 1 2}; do
  JB_DBG_SCHED=$m timeout -k 5 120 python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sched $m: step', round(d['ms_per_step'], 2), 'pg', round(d['ms_per_step'] - d['roofline']['kernel_ms'], 2), 'voc', round(d['roofline']['kernel_ms'], 2))"
done; done
