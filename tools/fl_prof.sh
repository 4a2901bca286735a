#!/bin/bash
# where a phase of the band solve goes: solver vs movers, work vs barrier wait (JB_FL_PROFILE build, restored on exit)
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_mlpg.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
(cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DJB_FL_PROFILE=1 $FLAGS -x hip -c jb_mlpg.hip -o build/jb_mlpg.o \
  && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
JB_ONE_STREAM=1 python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 2>&1 | grep "fl_pass\|build_mt2" | tail -${TAILN:-4}
