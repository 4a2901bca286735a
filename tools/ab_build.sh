#!/bin/bash
# A/B: frames per block of the [dim][frame] build kernel (LDS per block sets the waves per SIMD)
cd "$(dirname "$0")/.."
# whatever happens, leave the library built with the DEFAULT tile (build.sh only rebuilds stale objects)
trap 'rm -f jbonsai_amd/csrc/build/jb_mlpg.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
for tf in 32 16 24 64; do
  (cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DJB_BUILD_TF=$tf -x hip -c jb_mlpg.hip -o build/jb_mlpg.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
  echo "== JB_BUILD_TF=$tf"
  JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh 2>/dev/null | grep "build_mt"
  python bench.py --no-cpu-baseline --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', round(d['ms_per_step'],2))"
done
