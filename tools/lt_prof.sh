#!/bin/bash
# k_vocoder_lt: cycles one wave spends in its frame set-ups and in its sample loops (JB_LT_PROFILE build, restored on exit)
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_vocoder.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
(cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DJB_LT_PROFILE=1 $FLAGS -x hip -c jb_vocoder.hip -o build/jb_vocoder.o \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 2>&1 | grep "k_vocoder_lt wave" | tail -2
