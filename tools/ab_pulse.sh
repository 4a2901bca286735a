#!/bin/bash
# A/B: pulse scheduler as a work queue with N waves (0 = one lane per run, static)
cd "$(dirname "$0")/.."
for q in 0 128 256 512 1024 0 512; do
  echo "== JB_PULSE_QUEUE=$q"
  JB_PULSE_QUEUE=$q python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['ms_per_step']-d['roofline']['kernel_ms'])"
done
JB_ONE_STREAM=1 JB_PULSE_QUEUE=512 tools/kstats.sh 2>/dev/null | grep "pulse\|run_scan"
