#!/bin/bash
# A/B: vocoder warm-up length (frames each chunk starts early from zero state)
cd "$(dirname "$0")/.."
for w in 32 24 20 16 12; do
  echo "== JB_WARMUP_FRAMES=$w"
  JB_WARMUP_FRAMES=$w python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['chunks_redone_last_step'], d['config']['vocoder_work_items'])"
done
