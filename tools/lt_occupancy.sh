#!/bin/bash
# Where the dominant kernel's SIMDs hold fewer than two waves (VERDICT r5 "next" 6): the library built with
# -DJB_LT_STAMPS (tools/build_variant.sh stamps -DJB_LT_STAMPS) stamps every wave's start and end with the 100 MHz
# clock and its SIMD; tools/lt_occupancy.py turns the stamps of the LAST launch into the occupancy account.
cd "$(dirname "$0")/.."
lib=tools/_ab_stamps/libjbonsai_amd.so
[ -f $lib ] || { echo "build it first: tools/build_variant.sh stamps -DJB_LT_STAMPS"; exit 2; }
mkdir -p gpurun_out
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep_st.so
trap 'cp /tmp/_keep_st.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp $lib jbonsai_amd/libjbonsai_amd.so
JB_LT_STAMPS_FILE=gpurun_out/lt_stamps.txt timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $BENCH_ARGS 2>&1 | grep -a "per XCD\|ms_per_step" | cut -c1-160
python3 tools/lt_occupancy.py gpurun_out/lt_stamps.txt
