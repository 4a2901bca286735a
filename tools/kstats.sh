#!/bin/bash
# rocprofv3 kernel trace of one default bench run; prints per-kernel totals (ms) for the timed steps.
cd "$(dirname "$0")/.."
# (refuse --gpus: bench.py would become a launcher that starts its ranks from a process the profiler has
#  already initialised the GPU in -- the hop behind `--` that must not happen on this pool; profile one rank)
case " $BENCH_ARGS $* " in *" --gpus "*) echo "profile a single rank: no --gpus under rocprofv3"; exit 2;; esac
export TMPDIR=/tmp
out=gpurun_out/kstats
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o k --output-format csv -- python3 bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-extras "$@" > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log | cut -c1-200
python3 - $out ${STEPS:-2} <<'PY'
import sys,glob,csv,collections
d=sys.argv[1]; steps=int(sys.argv[2])+1
f=glob.glob(d+"/**/k_kernel_trace.csv",recursive=True)[0]
tot=collections.defaultdict(float); cnt=collections.Counter()
rows=list(csv.DictReader(open(f)))
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    tot[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; cnt[k]+=1
for k,v in sorted(tot.items(),key=lambda x:-x[1])[:24]:
    print(f"{v/steps:9.3f} ms/step  {cnt[k]/steps:6.1f} launches/step  {k[:90]}")
PY
