#!/bin/bash
# A/B: tile / block size of the time-parallel GV sweeps
cd "$(dirname "$0")/.."
IFS=";" read -ra LIST <<< "${CFGS:-2048 512;2048 1024;4096 1024;4096 512}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  (cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && HIPCC="/opt/rocm/bin/hipcc -DJB_GV_TT=$1 -DJB_GV_NT=$2" ./build.sh >/dev/null 2>&1)
  echo "== TT=$1 NT=$2"
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gv or mlpg" 2>&1 | tail -1
  JB_ONE_STREAM=1 tools/kstats.sh 2>/dev/null | grep "gv_tp"
done
(cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && ./build.sh >/dev/null 2>&1)
