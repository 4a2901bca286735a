#!/bin/bash
# config 2 as copies of the utterance of seeds 0..3, at 18 and 14 frames of chunk warm-up: ms per step, chunks redone
cd "$(dirname "$0")/.."
for seed in 0 1 2 3; do for wf in 18 14; do
  timeout -k 5 120 python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 --seed $seed --warmup-frames $wf 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('seed $seed warm-up $wf frames: step', round(d['ms_per_step'], 2), 'ms, vocoder kernel', round(d['roofline']['kernel_ms'], 2), 'ms, chunks redone', c['chunks_redone_last_step'], 'settled at checkpoint', c['chunks_settled_at_checkpoint_last_step'])"
done; done
