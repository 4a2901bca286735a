#!/bin/bash
# The clock the chip holds under the dominant kernel (GPU box): the library built with -DJB_LT_CLOCKS
# (tools/build_variant.sh clk -DJB_LT_CLOCKS) prints shader-clock ticks against the constant 100 MHz clock for two
# waves of k_vocoder_lt; tools/microbench/f64_clock.hip does the same for bare v_fma_f64 loops of the same length.
cd "$(dirname "$0")/.."
lib=tools/_ab_clk/libjbonsai_amd.so
[ -f $lib ] || { echo "build it first: tools/build_variant.sh clk -DJB_LT_CLOCKS"; exit 2; }
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep.so
trap 'cp /tmp/_keep.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp $lib jbonsai_amd/libjbonsai_amd.so
timeout -k 5 120 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $BENCH_ARGS 2>&1 | grep -a "k_vocoder_lt wave\|ms_per_step" | tail -8
[ -x tools/_ab_clk/f64_clock ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_ab_clk/f64_clock tools/microbench/f64_clock.hip
timeout -k 5 120 tools/_ab_clk/f64_clock
