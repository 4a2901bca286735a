#!/bin/bash
# where the resident GV kernel spends its time: -DJB_GG_PROFILE build, one-stream run (VARIANT = extra -D flags)
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_gv_gang.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
(cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DJB_GG_PROFILE=1 $VARIANT -x hip -c jb_gv_gang.hip -o build/jb_gv_gang.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
JB_ONE_STREAM=1 JB_GG_PROFILE_PRINT=1 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>&1 | grep "ticks" | tail -1
