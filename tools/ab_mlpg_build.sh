#!/bin/bash
# A/B of build variants of jb_mlpg.hip (VARIANTS = ';'-separated sets of -D flags): one-stream kernel times
# matching KERNELS (grep pattern) and the step; the normal build is restored on exit
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_mlpg.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
IFS=';' read -ra VS <<< "${VARIANTS:--DJB_BUILD_TF=32}"
for v in "${VS[@]}"; do
  (cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $v -x hip -c jb_mlpg.hip -o build/jb_mlpg.o 2>/dev/null \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
  echo "== $v"
  JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh 2>&1 | grep "${KERNELS:-build_mt}\|rror:"
  [ -n "$NOSTEP" ] || python bench.py --no-cpu-baseline --no-extras --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', round(d['ms_per_step'],2), 'pg', round(d['ms_per_step']-d['roofline']['kernel_ms'],2))"
done
