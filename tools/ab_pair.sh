#!/bin/bash
# A/B: lane-triple vs lane-pair throughput kernel
cd "$(dirname "$0")/.."
for k in triple pair; do
  echo "== JB_LP_KERNEL=$k"
  JB_LP_KERNEL=$k python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
  JB_LP_KERNEL=$k python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['chunks_redone_last_step'], d['config']['vocoder_work_items'])"
done
