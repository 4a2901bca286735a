#!/bin/bash
# shortest chunk of the lane-triple kernel for batches that do not fill the chip
cd "$(dirname "$0")/.."
for bsz in 8 16 32 64 128; do for cm in 48 24 16 8; do
  JB_CHUNK_MIN=$cm python bench.py --no-cpu-baseline --steps 5 --batch $bsz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print($bsz, $cm, d['ms_per_step'], d['roofline']['kernel_ms'], c['vocoder_work_items'], c['vocoder_chunk_frames'], c['chunks_redone_last_step'])"
done; done
