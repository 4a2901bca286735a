#!/bin/bash
# Mid-size batches of distinct utterances by vocoder kernel and chunk length (GPU box): the library's own choice,
# then the lane-triple kernel with its own and with shorter chunks (bench.py --kernel / --chunk-frames).
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --distinct 64 --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*: %.2f ms | %s chunk %d items %d | redone %d (%d settle) voc %.2f' % (d['ms_per_step'], d['roofline']['kernel'], c['vocoder_chunk_frames'], c['vocoder_work_items'], c['chunks_redone_last_step'], c['chunks_settled_at_checkpoint_last_step'], d['roofline']['kernel_ms']))"; }
for spec in "64 2000" "48 2000" "100 2000"; do set -- $spec
 run --batch $1 --frames $2
 run --batch $1 --frames $2 --kernel triple
 run --batch $1 --frames $2 --kernel triple --chunk-frames 12
 run --batch $1 --frames $2 --kernel triple --chunk-frames 8
done
