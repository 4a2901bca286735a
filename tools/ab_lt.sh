#!/bin/bash
# same-box A/B of build variants of jb_vocoder.hip (VARIANTS = ';'-separated sets of -D flags): vocoder kernel ms, step ms
cd "$(dirname "$0")/.."
trap 'rm -f jbonsai_amd/csrc/build/jb_vocoder.o && bash jbonsai_amd/csrc/build.sh > /dev/null' EXIT
IFS=';' read -ra VS <<< "${VARIANTS:--DJB_LT_CHUNKS=21;-DJB_LT_CHUNKS=20}"
for rep in 1 2; do
for v in "${VS[@]}"; do
  (cd jbonsai_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $v -x hip -c jb_vocoder.hip -o build/jb_vocoder.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libjbonsai_amd.so build/*.o) || exit 1
  python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step', round(d['ms_per_step'],2), 'voc', round(d['roofline']['kernel_ms'],2), 'chunk', d['config']['vocoder_chunk_frames'], 'items', d['config']['vocoder_work_items'])"
done
done
