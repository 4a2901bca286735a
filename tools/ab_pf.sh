#!/bin/bash
# A/B of k_postfilter build variants: each line of VARIANTS is a set of -D flags.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
while read -r flags; do
  [ -z "$flags" ] && continue
  ( cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -x hip -c jb_postfilter.hip -o build/jb_postfilter.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o ) || exit 1
  echo "== $flags"
  STEPS=1 bash tools/kstats.sh --beta 0.3 2>&1 | grep "k_postfilter"
done <<< "$VARIANTS"
