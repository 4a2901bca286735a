#!/bin/bash
# A/B: LDS prefetch depth of the lane-triple kernel
cd "$(dirname "$0")/.."
for pf in 2 3 4; do
  (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && HIPCC="/opt/rocm/bin/hipcc -DJB_LT_PF=$pf" ./build.sh >/dev/null 2>&1)
  echo "== JB_LT_PF=$pf"
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['chunks_redone_last_step'])"
done
