#!/bin/bash
# Everything profiles/rNN_* is made from, in one go on the GPU box (separate rocprofv3 passes):
#   bench line, kernel stats (streams overlapped and one-stream), timeline of one step, SQ counters of the
#   dominant kernel, HBM traffic per kernel, section profile of the resident GV kernel.
# usage: bash tools/profile_round.sh r03      -> gpurun_out/prof_r03/* (and the files bench.py quotes -> profiles/)
cd "$(dirname "$0")/.."
R=${1:-r03}
out=gpurun_out/prof_$R
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
# counters first: the bench line quotes them (profiles/${R}_traffic.json, profiles/${R}_pmc_sq_k_vocoder_lt.json;
# copy them from $out to profiles/ after the call: only gpurun_out/ comes back from the box)
JSON=profiles/${R}_pmc_sq_k_vocoder_lt.json bash tools/pmc_voc.sh > $out/${R}_pmc_sq_k_vocoder_lt.txt 2>&1 && cp profiles/${R}_pmc_sq_k_vocoder_lt.json $out/
bash tools/traffic.sh $R > $out/traffic.log 2>&1 && cp profiles/${R}_traffic.json $out/${R}_traffic.json
python bench.py --steps 20 --warmup 5 > $out/${R}_bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
STEPS=4 bash tools/kstats.sh > $out/${R}_kernel_ms_per_step.txt 2>&1
cp gpurun_out/kstats/k_kernel_stats.csv $out/${R}_kernel_stats.csv
python tools/timeline.py > $out/${R}_timeline_one_step.txt
JB_ONE_STREAM=1 STEPS=4 bash tools/kstats.sh > $out/${R}_kernel_ms_per_step_one_stream.txt 2>&1
KERNEL=k_mlpg_gv_gang bash tools/pmc_voc.sh > $out/${R}_pmc_sq_k_mlpg_gv_gang.txt 2>&1
bash tools/gg_prof.sh > $out/${R}_gv_gang_sections.txt 2>&1
bash tools/shapes.sh > $out/${R}_shapes.txt 2>&1
bash tools/small_shapes.sh > $out/${R}_small_shapes.txt 2>&1
python tools/power_trace.py > $out/${R}_power.txt 2>&1
bash tools/seed_sweep.sh > $out/${R}_seed_sweep.txt 2>&1
python tools/latency_single.py > $out/${R}_latency_single.txt 2>&1
ls -la $out
