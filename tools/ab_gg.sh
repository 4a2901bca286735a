#!/bin/bash
# A/B of build variants of the resident GV kernel (VARIANTS = ';'-separated sets of -D flags), alone-time + step
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_gv_gang.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
IFS=';' read -ra VS <<< "${VARIANTS:--DJB_GG_XCHG=3;-DJB_GG_XCHG=2}"
for v in "${VS[@]}"; do
  (cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $v -x hip -c jb_gv_gang.hip -o build/jb_gv_gang.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
  echo "== $v"
  JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh 2>&1 | grep "gv_gang\|rror"
  python bench.py --no-cpu-baseline --no-extras --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', round(d['ms_per_step'],2), 'pg', round(d['ms_per_step']-d['roofline']['kernel_ms'],2))"
done
