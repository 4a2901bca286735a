#!/bin/bash
# HBM traffic per kernel for one bench step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# SEPARATE passes (they do not fit one), kernel trace only.  Writes profiles/${R}_traffic.json (R = $1,
# default r03) with the workload and the kernel-source fingerprint of the bench line it was measured on:
# bench.py quotes it only for that workload on those sources.
cd "$(dirname "$0")/.."
# (refuse --gpus: bench.py would become a launcher that starts its ranks from a process the profiler has
#  already initialised the GPU in -- the hop behind `--` that must not happen on this pool; profile one rank)
case " $BENCH_ARGS $* " in *" --gpus "*) echo "profile a single rank: no --gpus under rocprofv3"; exit 2;; esac
export TMPDIR=/tmp
R=${1:-r03}
out=gpurun_out/traffic
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $out/$c.log; exit 1; }
done
python3 - $out $R <<'PY'
import sys, glob, csv, json, collections
out, R = sys.argv[1], sys.argv[2]
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.Counter()
for c in acc:
    for fn in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[c][k] += float(r["Counter_Value"]) * 1024.0  # counters are in KB
            if c == "FETCH_SIZE":
                cnt[k] += 1
bench = json.loads([l for l in open(f"{out}/FETCH_SIZE.log").read().splitlines() if l.startswith('{"metric"')][-1])
cfg = bench["config"]
dom = max((k for k in acc["FETCH_SIZE"] if "k_vocoder" in k), key=lambda k: acc["WRITE_SIZE"][k] + acc["FETCH_SIZE"][k])
samples = cfg["samples_per_step_per_gpu"]
res = {
    "batch": cfg["batch_per_gpu"], "frames": cfg["frames_per_utterance"], "kernel": dom,
    "kernel_sources_sha16": bench.get("kernel_sources_sha16"),
    "whole_step_hbm_bytes": sum(acc["FETCH_SIZE"].values()) + sum(acc["WRITE_SIZE"].values()),
    "whole_step_fetch_bytes_raw": sum(acc["FETCH_SIZE"].values()), "whole_step_write_bytes": sum(acc["WRITE_SIZE"].values()),
    "fetch_bytes_raw": acc["FETCH_SIZE"][dom], "write_bytes": acc["WRITE_SIZE"][dom],
    "hbm_bytes_per_launch": acc["FETCH_SIZE"][dom] + acc["WRITE_SIZE"][dom],
    "algorithmic_bytes_per_launch": 8.67 * samples,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB x 1024), bench.py --steps 1 "
            "--warmup 0.  FETCH_SIZE is NOT doubled: the dominant kernel's loads are 8 B/lane (excitation) and "
            "scalar-sized coefficient reads, an access width MI355X_MICROARCH.md calls uncalibrated (the 2x "
            "correction is established only for 16 B/lane streams); WRITE_SIZE is exact for 16 B/lane streaming "
            "stores.  The kernel reads the excitation (8 B/sample, written by k_excite_w4/k_excite_fix) in addition "
            "to the per-frame coefficients, so its reads exceed the algorithmic input bytes by construction.",
    "all_kernels": {k: {"fetch_bytes_raw": acc["FETCH_SIZE"][k], "write_bytes": acc["WRITE_SIZE"][k], "launches": cnt[k]}
                    for k in sorted(acc["FETCH_SIZE"])},
}
json.dump(res, open(f"profiles/{R}_traffic.json", "w"), indent=1)
tot_f = sum(acc["FETCH_SIZE"].values()); tot_w = sum(acc["WRITE_SIZE"].values())
print(f"dominant {dom}: fetch {res['fetch_bytes_raw']/1e9:.2f} GB write {res['write_bytes']/1e9:.2f} GB; whole step fetch {tot_f/1e9:.1f} GB write {tot_w/1e9:.1f} GB")
PY
