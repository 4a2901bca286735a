#!/bin/bash
# The library's HOST code under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on this pool):
# builds tools/_ab_asan/libjbonsai_amd.so with -fsanitize=address,undefined for the host pass only, swaps it in for
# one run of the CPU test suite (front half: voice parsing, tree search, durations, label parsing, conditions, ABI
# structs, LPT partition, gather bookkeeping) and puts the product library back.
cd "$(dirname "$0")/.."
set -e
bash tools/build_variant.sh asan -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1 2>&1 | tail -1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep_asan.so
trap 'cp /tmp/_keep_asan.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp tools/_ab_asan/libjbonsai_amd.so jbonsai_amd/libjbonsai_amd.so
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_c_consumer.py "$@"   # (that test links a C program against the library with gcc: no sanitizer runtime there)
