#!/bin/bash
# kernel timeline of one step with some launches left out (JB_DBG_SKIP mask, -DJB_DBG_GATES library): tools/gate_timeline.sh MASK
cd "$(dirname "$0")/.."
cp jbonsai_amd/libjbonsai_amd.so /tmp/_keep.so
trap 'cp /tmp/_keep.so jbonsai_amd/libjbonsai_amd.so' EXIT
cp tools/_ab_gates/libjbonsai_amd.so jbonsai_amd/libjbonsai_amd.so
export JB_DBG_SKIP=$1 JB_DBG_SKIP_AFTER=1
STEPS=3 bash tools/kstats.sh > /dev/null 2>&1
python tools/timeline.py 2>&1 | head -${2:-32}
