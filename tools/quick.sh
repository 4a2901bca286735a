#!/bin/bash
# quick look on the GPU box: GV/parity tests, one-stream kernel times, step time
cd "$(dirname "$0")/.."
out=gpurun_out/quick; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh "$@" 2>&1 | grep -v "vocoder\|rocprof" | head -${TOP:-9}
python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['ms_per_step'],2), 'voc', round(d['roofline']['kernel_ms'],2), 'pg', round(d['ms_per_step']-d['roofline']['kernel_ms'],2))"
