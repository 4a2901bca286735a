#!/bin/bash
# A/B: per-tap FMAs of k_vocoder_lt as one asm block (1) or single-instruction asm statements (0)
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do
  (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && HIPCC="/opt/rocm/bin/hipcc -DJB_LT_MERGED=$v" ./build.sh >/dev/null 2>&1)
  echo "== JB_LT_MERGED=$v"
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['chunks_redone_last_step'])"
done
(cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && ./build.sh >/dev/null 2>&1)
