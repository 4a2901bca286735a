#!/bin/bash
# Bits of the tracks, the excitation and the PCM of fixed batches (tests/tools/ab_bits.py) under two prebuilt libraries:
#   tools/ab_bits.sh tools/_ab_base/libjbonsai_amd.so [other.so]      (second default: the product library)
cd "$(dirname "$0")/.."
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep_bits.so
trap 'cp /tmp/_keep_bits.so jbonsai_amd/libjbonsai_amd.so' EXIT
A=$1; B=${2:-/tmp/_keep_bits.so}
cp "$A" jbonsai_amd/libjbonsai_amd.so; python tests/tools/ab_bits.py > /tmp/_bits_a.txt 2>&1
cp "$B" jbonsai_amd/libjbonsai_amd.so; python tests/tools/ab_bits.py > /tmp/_bits_b.txt 2>&1
echo "== $A"; cat /tmp/_bits_a.txt; echo "== $B"; cat /tmp/_bits_b.txt
if diff -q /tmp/_bits_a.txt /tmp/_bits_b.txt > /dev/null; then echo "BITS IDENTICAL"; else echo "BITS DIFFER"; diff /tmp/_bits_a.txt /tmp/_bits_b.txt; fi
