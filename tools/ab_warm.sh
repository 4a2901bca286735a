#!/bin/bash
# vocoder warm-up length against the cost of the chunks it makes fail (distinct utterances: copies never fail)
cd "$(dirname "$0")/.."
for w in ${WARMS:-20 16 12}; do
  for d in 1 64; do
    JB_WARMUP_FRAMES=$w python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 --distinct $d 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warm-up $w distinct $d: step', round(d['ms_per_step'],2), 'voc', round(d['roofline']['kernel_ms'],2), {k:v for k,v in d.items() if 'redo' in k or 'chunk' in k})"
  done
done
