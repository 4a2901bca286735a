#!/bin/bash
# Builds the library of another git revision beside the product: tools/build_rev.sh REV [NAME]
#   -> tools/_ab_NAME/libjbonsai_amd.so (default NAME = base) for same-box A/B runs (tools/ab_libs.sh)
set -euo pipefail
cd "$(dirname "$0")/.."
rev=$1; name=${2:-base}
tmp=$(mktemp -d); trap 'rm -rf $tmp' EXIT
git archive "$rev" jbonsai_amd/csrc include | tar -x -C "$tmp"
out=$PWD/tools/_ab_$name; mkdir -p "$out"
(cd "$tmp/jbonsai_amd/csrc" && bash build.sh > /dev/null && cp ../libjbonsai_amd.so "$out/")
echo "built $out/libjbonsai_amd.so from $rev"
