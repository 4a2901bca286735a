#!/bin/bash
# VERDICT r5 "next" 1, the gate: wall times (tools/stagger_gate.py) + a kernel timeline of one staggered pass
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/stagger; rm -rf $out; mkdir -p $out
python3 tools/stagger_gate.py --groups 4 --delays 0,2,5,8 > $out/g4.txt 2>&1 || { tail -5 $out/g4.txt; exit 1; }
cat $out/g4.txt
if [ -z "$ONLY4" ]; then
python3 tools/stagger_gate.py --groups 8 --delays 0,2,4 > $out/g8.txt 2>&1; cat $out/g8.txt
python3 tools/stagger_gate.py --groups 2 --delays 0,5,10 > $out/g2.txt 2>&1 || { tail -5 $out/g2.txt; exit 1; }
cat $out/g2.txt
fi
rocprofv3 --kernel-trace -d $out/tr -o k --output-format csv -- python3 tools/stagger_gate.py --groups 4 --delays 5 --trace > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python3 - $out <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/tr/**/k_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the traced pass = everything from the 5th-from-last k_pitch on (4 groups: the last 4 k_pitch launches)
pitch = [i for i, r in enumerate(rows) if "k_pitch" in r["Kernel_Name"]]
lo = pitch[-4]
while lo > 0 and int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 2_000_000:
    lo -= 1
t0 = int(rows[lo]["Start_Timestamp"])
with open(sys.argv[1] + "/timeline_g4_5ms.txt", "w") as o:
    for r in rows[lo:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) < 150_000:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jb::", "")
        o.write(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  q{r.get('Queue_Id', '?'):>3}  {name[:60]}\n")
print(open(sys.argv[1] + "/timeline_g4_5ms.txt").read())
PY
