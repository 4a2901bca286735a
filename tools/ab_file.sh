#!/bin/bash
# same-box A/B of two versions of one source file of the library: tools/ab_file.sh jb_vocoder.hip fileA fileB [kernel pattern]
# (the file in the tree is restored and the library rebuilt on exit)
cd "$(dirname "$0")/.."
src=jbonsai_amd/csrc/$1; a=$2; b=$3; pat=${4:-vocoder_lt}
cp $src /tmp/_ab_keep
trap 'cp /tmp/_ab_keep '"$src"' && touch '"$src"' && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
for round in 1 2; do
for v in $a $b; do
  cp $v $src && touch $src && bash jbonsai_amd/csrc/build.sh > /dev/null 2>&1 || exit 1
  echo "== $v"
  JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh --no-extras 2>&1 | grep "$pat"
  python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', round(d['ms_per_step'],2), 'pg', round(d['ms_per_step']-d['roofline']['kernel_ms'],2))"
done; done
