#!/bin/bash
# Same-box A/B of versions of ONE source file of the library (run on the GPU box, from anywhere):
#   tools/ab.sh [-f jb_vocoder.hip] [-k 'kernel grep pattern'] [-r rounds] [-b 'bench.py args'] VARIANT...
# A VARIANT is either a set of -D flags for the file ("-DJB_X=2 -DJB_Y"), or file:PATH = another version of
# the file (e.g. a copy made locally with `git show REV:jbonsai_amd/csrc/<file>`; the GPU box has no .git),
# or "base" = the file as it is.  For every variant: the one-stream (alone) times of the kernels matching
# the pattern, then the step and its parameter-generation part.  The file in the tree is restored and the
# library rebuilt on exit.  (This replaces the per-experiment ab_*.sh scripts of rounds 1-2.)
cd "$(dirname "$0")/.."
file=jb_vocoder.hip; pat=vocoder_lt; rounds=1; bargs=""
while getopts "f:k:r:b:" o; do
  case $o in f) file=$OPTARG;; k) pat=$OPTARG;; r) rounds=$OPTARG;; b) bargs=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { sed -n 2,10p "$0"; exit 2; }
src=jbonsai_amd/csrc/$file; obj=jbonsai_amd/csrc/build/${file%.*}.o
cp "$src" /tmp/_ab_keep
trap 'cp /tmp/_ab_keep '"$src"' && rm -f '"$obj"' && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
for round in $(seq 1 "$rounds"); do
for v in "$@"; do
  cp /tmp/_ab_keep "$src"; defs=""
  case "$v" in file:*) cp "${v#file:}" "$src";; base) ;; *) defs=$v;; esac
  (cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c "$file" -o "build/${file%.*}.o" 2> /tmp/_ab_err \
     && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || { tail -5 /tmp/_ab_err; exit 1; }
  echo "== $v"
  JB_ONE_STREAM=1 STEPS=2 bash tools/kstats.sh --no-extras 2>&1 | grep -E "$pat|rror"
  python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $bargs 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step', round(d['ms_per_step'], 2), 'ms, parameter generation + excitation', round(d['ms_per_step'] - d['roofline']['kernel_ms'], 2),
      'ms, vocoder kernel', round(d['roofline']['kernel_ms'], 2), 'ms, chunks redone', d['config']['chunks_redone_last_step'])"
done; done
