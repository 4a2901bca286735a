"""GPU busy / idle of a rocprofv3 --kernel-trace run: union of kernel intervals over all queues, the gaps
between them above a threshold, and what ran around each gap.  usage: gaps.py DIR [min_gap_ms] [last_seconds]"""
import csv
import glob
import sys

d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jb::", ""))
        for r in csv.DictReader(open(f))]
rows.sort()
if len(sys.argv) > 3:
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - float(sys.argv[3]) * 1e9]
t0 = rows[0][0]
busy, cur_s, cur_e, last_name, gaps = 0, rows[0][0], rows[0][1], rows[0][2], []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if (s - cur_e) / 1e6 >= thr:
            gaps.append(((cur_e - t0) / 1e6, (s - cur_e) / 1e6, last_name, n))
        cur_s, cur_e, last_name = s, e, n
    elif e > cur_e:
        cur_e, last_name = e, n
busy += cur_e - cur_s
span = (cur_e - t0) / 1e6
print(f"span {span:.1f} ms, GPU busy {busy / 1e6:.1f} ms ({100 * busy / 1e6 / span:.1f} %), idle {span - busy / 1e6:.1f} ms")
for at, g, a, b in gaps:
    print(f"  at {at:9.1f} ms: idle {g:6.2f} ms   after {a[:40]:40s} before {b[:40]}")
