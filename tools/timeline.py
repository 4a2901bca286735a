"""Kernel timeline of the LAST step of a kstats.sh run (gpurun_out/kstats/k_kernel_trace.csv):
start offset, duration and queue of every kernel, in start order -- what overlaps what, and
where the vocoder has to wait."""
import csv
import glob
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kstats"
f = glob.glob(d + "/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with its first k_prep_states* launch (one k_pitch per step tells the steps apart)
pitch = [i for i, r in enumerate(rows) if "k_pitch" in r["Kernel_Name"]]
if len(pitch) < 2:
    raise SystemExit("need two steps in the trace")
lo = next(i for i in range(pitch[-2] + 1, len(rows)) if "k_prep_states" in rows[i]["Kernel_Name"])
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jb::", "")
    print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  q{r.get('Queue_Id', '?'):>3}  {name[:60]}")
