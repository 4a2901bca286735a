#!/bin/bash
# Counters of EVERY kernel of one step against its alone-time (JB_ONE_STREAM), one rocprofv3 pass per set:
# which unit could bound a kernel?  scalar ALU: one per CU (256); VALU: one per SIMD (1024); LDS: one per CU;
# then the memory side: TLB misses, L2 hit rate, mean latency of a vector-L1 read request, address-unit busy.
cd "$(dirname "$0")/.."
# (refuse --gpus: bench.py would become a launcher that starts its ranks from a process the profiler has
#  already initialised the GPU in -- the hop behind `--` that must not happen on this pool; profile one rank)
case " $BENCH_ARGS $* " in *" --gpus "*) echo "profile a single rank: no --gpus under rocprofv3"; exit 2;; esac
export TMPDIR=/tmp JB_ONE_STREAM=1
out=gpurun_out/pmc_all
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INST_CYCLES_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 - $out <<'PY'
import sys, glob, csv, collections
d = sys.argv[1]
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jb::", "")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[name(r)][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float)
for fn in glob.glob(d + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[name(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"{'kernel':26s} {'ms':>6s} {'SALU/CU':>8s} {'VALU/SIMD':>9s} {'LDS/CU':>7s} {'TLBmiss%':>8s} {'L2hit%':>6s} {'rdlat':>6s} {'TAbusy%':>7s} {'waves':>8s}"
      "   (M instructions; at 2.1 GHz 1 ms = 2.1 M cycles; rdlat = cycles per L1->L2 read request)")
for k, v in sorted(dur.items(), key=lambda x: -x[1])[:16]:
    a = acc[k]
    tl = a['TCP_UTCL1_TRANSLATION_MISS_sum'] + a['TCP_UTCL1_TRANSLATION_HIT_sum']
    l2 = a['TCC_HIT_sum'] + a['TCC_MISS_sum']
    rq = a['TCP_TCC_READ_REQ_sum']
    print(f"{k[:26]:26s} {v:6.2f} {a['SQ_INSTS_SALU'] / 256e6:8.2f} {a['SQ_INSTS_VALU'] / 1024e6:9.2f} {a['SQ_INSTS_LDS'] / 256e6:7.2f} "
          f"{100 * a['TCP_UTCL1_TRANSLATION_MISS_sum'] / tl if tl else 0:8.2f} {100 * a['TCC_HIT_sum'] / l2 if l2 else 0:6.1f} "
          f"{a['TCP_TCC_READ_REQ_LATENCY_sum'] / rq if rq else 0:6.0f} {a['TA_BUSY_avr']:7.1f} {a['SQ_WAVES']:8.0f}")
PY
