#!/bin/bash
# diagnosis of k_mlpg_fb_lds: 0 = real, 1 = movers only (solver idle), 2 = solver only (movers idle; results wrong)
cd "$(dirname "$0")/.."
for v in 0 1 2; do
  (cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && HIPCC="/opt/rocm/bin/hipcc -DJB_FL_TEST=$v" ./build.sh >/dev/null 2>&1)
  echo "== JB_FL_TEST=$v"
  JB_ONE_STREAM=1 tools/kstats.sh 2>/dev/null | grep "fb_lds"
done
(cd jbonsai_amd/csrc && rm -f build/jb_mlpg.o && ./build.sh >/dev/null 2>&1)
