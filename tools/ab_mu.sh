#!/bin/bash
# A/B: chunk size of the MLPG solver (latency- vs compute-bound?)
set -e
cd "$(dirname "$0")/.."
for mu in 4 8 12; do
  (cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && HIPCC="/opt/rocm/bin/hipcc -DJB_MU=$mu" ./build.sh >/dev/null)
  echo "== MU=$mu"
  python tests/tools/probe_time.py 2000 256 2>/dev/null | grep "B=256" | tail -1
done
