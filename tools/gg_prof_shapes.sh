#!/bin/bash
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_gv_gang.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
(cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DJB_GG_PROFILE=1 -x hip -c jb_gv_gang.hip -o build/jb_gv_gang.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
for spec in "1024 3900" "512 7800" "256 25546"; do set -- $spec
echo "== batch $1 frames $2"
JB_ONE_STREAM=1 JB_GG_PROFILE_PRINT=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch $1 --frames $2 2>&1 | grep "ticks" | tail -1
done
