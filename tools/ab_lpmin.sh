#!/bin/bash
# where does the lane-triple kernel start to beat the wave kernel?  (frames in the batch)
cd "$(dirname "$0")/.."
for bsz in 4 8 12 16 32; do for lp in 100000000 1; do
  JB_LP_MIN_FRAMES=$lp python bench.py --no-cpu-baseline --steps 5 --batch $bsz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print($bsz, d['roofline']['kernel'] if False else ('lt' if $lp==1 else 'wave'), d['ms_per_step'], d['roofline']['kernel_ms'], c['vocoder_work_items'], c['vocoder_chunk_frames'])"
done; done
