#!/bin/bash
# SQ counter passes for the dominant vocoder kernel (one bench step each; separate passes).  With JSON=path
# the totals are also written as a record carrying the workload and the kernel-source fingerprint of the
# bench line they were measured on (profiles/r03_pmc_sq_k_vocoder_lt.json: what bench.py may quote).
cd "$(dirname "$0")/.."
# (refuse --gpus: bench.py would become a launcher that starts its ranks from a process the profiler has
#  already initialised the GPU in -- the hop behind `--` that must not happen on this pool; profile one rank)
case " $BENCH_ARGS $* " in *" --gpus "*) echo "profile a single rank: no --gpus under rocprofv3"; exit 2;; esac
export TMPDIR=/tmp
out=gpurun_out/pmc_voc
mkdir -p $out
[ -n "$JSON" ] && rm -f "$JSON"
export JSON
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  python3 - "$out/p$i" "${KERNEL:-k_vocoder_l}" <<'PY'
import sys,glob,csv,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(float); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]
        if (sys.argv[2] if len(sys.argv) > 2 else "k_vocoder_l") in k:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k,v in acc.items(): print(f"{k} {v:.4g} (dispatches {n[k]})")
import json, os
jp = os.environ.get("JSON")
if jp:
    rec = json.load(open(jp)) if os.path.exists(jp) else {"counters": {}}
    rec["counters"].update({k: v for k, v in acc.items()})
    line = [l for l in open(sys.argv[1] + ".log").read().splitlines() if l.startswith('{"metric"')][-1]
    b = json.loads(line)
    rec.update(kernel=sys.argv[2], batch=b["config"]["batch_per_gpu"], frames=b["config"]["frames_per_utterance"],
               kernel_sources_sha16=b.get("kernel_sources_sha16"),
               note="rocprofv3 --pmc, four separate passes of `bench.py --steps 1 --warmup 0`, totals over the kernel's dispatches")
    json.dump(rec, open(jp, "w"), indent=1)
PY
done
