#!/bin/bash
# A/B: s_setprio(3) in the LF0-chain kernels (k_mlpg_gv_vt, k_pulse) on/off
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do
  (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o build/jb_mlpg.o && HIPCC="/opt/rocm/bin/hipcc -DJB_SIDE_PRIO=$v" ./build.sh >/dev/null 2>&1)
  echo "== JB_SIDE_PRIO=$v"
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['ms_per_step']-d['roofline']['kernel_ms'])"
done
(cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o build/jb_mlpg.o && ./build.sh >/dev/null 2>&1)
