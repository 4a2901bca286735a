#!/bin/bash
# A/B: vocoder occupancy (items target) x batches in flight
cd "$(dirname "$0")/.."
for tgt in 40960 20480; do for pl in 1 2; do
  echo "== JB_LP_TARGET=$tgt pipeline=$pl"
  JB_LP_TARGET=$tgt python bench.py --steps 4 --warmup 2 --no-cpu-baseline --pipeline $pl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['chunks_redone_last_step'], d['config']['vocoder_work_items'], d['config']['vocoder_chunk_frames'])"
done; done
