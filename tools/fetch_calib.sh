#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against KNOWN byte counts per access width (tools/microbench/fetch_calib.hip), two
# rocprofv3 --pmc passes.  Prints reported / known per kernel: the factor tools/traffic.sh applies.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/fetch_calib
rm -rf $out; mkdir -p $out
[ -x tools/microbench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c -o p --output-format csv -- tools/microbench/fetch_calib > $out/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $out/$c.log; exit 1; }
done
python3 - $out <<'PY'
import sys, glob, csv, re
out = sys.argv[1]
log = open(f"{out}/FETCH_SIZE.log").read()
known = {"default": int(re.search(r"stream kernels known bytes (\d+)", log).group(1))}
for k in ("rd8_rows", "rd8_lane_stream"):
    known[k] = int(re.search(rf"{k} known bytes (\d+)", log).group(1))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("__amd"):
                continue
            rep = float(r["Counter_Value"]) * 1024.0
            kb = known.get(k.split("<")[0], known["default"])
            if (c == "FETCH_SIZE") == k.startswith("rd"):
                print(f"{c:10s} {k:28s} reported {rep / 1e9:7.3f} GB  known {kb / 1e9:7.3f} GB  reported/known {rep / kb:5.3f}")
PY
