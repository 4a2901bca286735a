#!/bin/bash
# A/B: frames per iteration (64 x NB) of the LF0 GV kernel
cd "$(dirname "$0")/.."
for nb in 4 8 16; do
  (cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && HIPCC="/opt/rocm/bin/hipcc -DJB_VT_NB=$nb" ./build.sh >/dev/null 2>&1)
  echo "== JB_VT_NB=$nb"
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mlpg or bonsai" 2>&1 | tail -1
  JB_ONE_STREAM=1 tools/kstats.sh 2>/dev/null | grep "gv_vt"
done
(cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && ./build.sh >/dev/null 2>&1)
