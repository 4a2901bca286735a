#!/bin/bash
# A/B: scheduling-barrier spacing in the lane-pair kernel (LDS read-ahead vs register pressure)
cd "$(dirname "$0")/.."
for sb in 1 2 3 4 6; do
  (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && HIPCC="/opt/rocm/bin/hipcc -DJB_LP_SB=$sb" ./build.sh >/dev/null 2>&1)
  echo "== JB_LP_SB=$sb"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --pipeline 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
