#!/bin/bash
# The library's host threads (per-label loops of the front half, JB_HOST_THREADS) under ThreadSanitizer on the CPU:
# builds tools/_ab_tsan/libjbonsai_amd.so with -fsanitize=thread for the host pass, swaps it in for the front-half,
# ABI and two-voice tests and puts the product library back.  (The read-back finisher and the staging ring need a GPU.)
cd "$(dirname "$0")/.."
set -e
bash tools/build_variant.sh tsan -fsanitize=thread -fno-gpu-sanitize -g -O1 2>&1 | tail -1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep_tsan.so
trap 'cp /tmp/_keep_tsan.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp tools/_ab_tsan/libjbonsai_amd.so jbonsai_amd/libjbonsai_amd.so
LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 python -m pytest tests/test_host_frontend.py tests/test_abi.py \
  tests/test_two_voices.py tests/test_host_fuzz.py -q -s -p no:cacheprovider "$@"
