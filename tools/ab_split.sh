#!/bin/bash
# CU partition sweep: two batches in flight, k CUs per XCD for parameter generation
cd "$(dirname "$0")/.."
python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline 1            ', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['vocoder_work_items'])"
for k in ${SPLITS:-0 6 8 10 12}; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --pipeline 2 --cu-split $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline 2 cu-split $k', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['vocoder_work_items'], d['config']['chunks_redone_last_step'])"
done
