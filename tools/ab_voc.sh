#!/bin/bash
# A/B ablations of k_vocoder (diagnostic builds; outputs are wrong by design)
set -e
cd "$(dirname "$0")/.."
for def in "" "-DJB_ABL_NO_PHASE_A" "-DJB_ABL_NO_PHASE_B"; do
  (cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && HIPCC="/opt/rocm/bin/hipcc $def" ./build.sh >/dev/null)
  echo "== build [$def]"
  python tests/tools/probe_time.py 2000 256 2>/dev/null | grep "B=256" | tail -1
done
(cd jbonsai_amd/csrc && rm -f build/jb_vocoder.o && ./build.sh >/dev/null)
