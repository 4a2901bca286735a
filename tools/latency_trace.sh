#!/bin/bash
# kernels of ONE warm single-sentence synthesis (SAMPLE_SENTENCE_1, 277 frames): rocprofv3 kernel trace of
# tools/latency_trace.py, the last call's launches with start offsets and durations; JB_E2E_TIMING phases beside it
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/lat
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python3 tools/latency_trace.py > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - $out <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last call: kernels after the last gap > 300 us
starts = [int(r["Start_Timestamp"]) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - int(rows[i - 1]["End_Timestamp"]) > 300000:
        cut = i
last = rows[cut:]
t0 = int(last[0]["Start_Timestamp"])
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  + {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]}")
print(f"{len(last)} launches, span {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, sum of kernel times {busy / 1e3:.1f} us")
PY
JB_E2E_TIMING=1 python3 tools/latency_trace.py 2>&1 | tail -12
