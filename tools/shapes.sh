#!/bin/bash
# the shapes table of DESIGN.md section 5 (profiles/rNN_shapes.txt): other batch shapes with DISTINCT utterances, one box
cd "$(dirname "$0")/.."
for spec in "1 500" "16 2000" "64 2000" "256 2000" "1024 500" "16 25546" "64 25546" "256 25546"; do
  set -- $spec
  python bench.py --batch $1 --frames $2 --distinct 64 --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('| $1 x $2 | %.1f | %.2f | %d | %s, %d, %d | %d (%d settle at the checkpoint) |' % (d['ms_per_step'], d['value']/1e9, d['realtime_factor'], d['roofline']['kernel'], c['vocoder_chunk_frames'], c['vocoder_work_items'], c['chunks_redone_last_step'], c['chunks_settled_at_checkpoint_last_step']))"
done
python bench.py --mixed --batch 512 --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('mixed 512: %.1f ms/step %.2f Gsamples/s voc %.1f redo %d' % (d['ms_per_step'], d['value']/1e9, d['roofline']['kernel_ms'], c['chunks_redone_last_step']))"
python bench.py --pipeline 2 --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline 2: %.1f ms/step' % d['ms_per_step'])"
