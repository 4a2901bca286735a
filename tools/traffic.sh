#!/bin/bash
# HBM traffic per kernel for one bench step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# SEPARATE passes (they do not fit one), kernel trace only.  Writes profiles/${R}_traffic.json (R = $1)
# with the workload and the kernel-source fingerprint of the bench line it was measured on: bench.py quotes
# it only for that workload on those sources.
# FETCH_SIZE on gfx950 tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, "HBM"): the raw figure is
# CALIBRATED on this box, in this call, per access shape (tools/fetch_calib.sh: known byte counts read 4 / 8 / 16 B
# per lane, as 128-byte pieces of far-apart rows, and as one stream per lane), and every kernel's reads are
# checked against the bytes it MUST read (the array inventory of DESIGN.md section 3).
cd "$(dirname "$0")/.."
# (refuse --gpus: bench.py would become a launcher that starts its ranks from a process the profiler has
#  already initialised the GPU in -- the hop behind `--` that must not happen on this pool; profile one rank)
case " $BENCH_ARGS $* " in *" --gpus "*) echo "profile a single rank: no --gpus under rocprofv3"; exit 2;; esac
export TMPDIR=/tmp
R=${1:-r05}
out=gpurun_out/traffic
rm -rf $out; mkdir -p $out
bash tools/fetch_calib.sh > $out/fetch_calibration.txt 2>&1 || { echo "calibration failed"; tail -5 $out/fetch_calibration.txt; exit 1; }
cat $out/fetch_calibration.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $out/$c.log; exit 1; }
done
python3 - $out $R <<'PY'
import sys, glob, csv, json, collections, re
out, R = sys.argv[1], sys.argv[2]
# ---- calibration: reported / known per access shape
cal = {}
for ln in open(f"{out}/fetch_calibration.txt"):
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(\S+)\s+reported.*reported/known\s+([\d.]+)", ln)
    if m:
        cal[m.group(2)] = float(m.group(3))
# Every coalesced shape measured -- 4, 8 and 16 B per lane, and 128-byte pieces of rows 200 KB apart -- reports exactly
# half its bytes (requests of 128 B tallied at 64 B), so the raw figure is divided by that for EVERY kernel.  The
# lane-stream shape (one stream per lane, 8 B per access: the vocoder's excitation reads) is not a counter calibration
# but an over-fetch measurement -- the microbenchmark's lanes evict each other's lines and fetch 8x their bytes -- and
# is kept in the file as such; the vocoder kernel's own figure is given with its lower bound (the raw tally).
RD8 = "rd<HIP_vector_type<unsigned int, 2u> >"
shape_of = {  # how a kernel reads its bulk (jb_mlpg.hip, jb_gv_gang.hip, jb_vocoder.hip)
    "k_mlpg_fb_lds": "rd8_rows", "k_mlpg_gv_gang": "rd8_rows", "k_mc2b_mt": "rd8_rows", "k_mlpg_fb_runs": "rd8_rows",
}
def shape(k):
    base = k.replace("jb::", "").split("<")[0]
    return shape_of.get(base, RD8)
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
B, T = cfg["batch_per_gpu"], cfg["frames_per_utterance"]
samples = cfg["samples_per_step_per_gpu"]
L = 35
arr = B * L * ((T + 15) // 16 * 16) * 8.0  # one [dim][frame] f64 workspace array of the MCP stream, rows on 128-byte lines
must_read = {  # compulsory reads per launch, from the array inventory
    "k_mlpg_fb_lds": (7 * arr, "forward A0 A1 A2 b + backward L1 L2 g/D"),
    "k_mlpg_gv_gang": (5 * arr, "A0 A1 A2 b par once (the 24 halo frames of every 512-frame window are read twice: x1.049)"),
    "k_mc2b_mt": (1 * arr, "par"),
}
must_write = {
    "k_mlpg_build_mt2": (4 * arr, "A0 A1 A2 b"), "k_mlpg_fb_lds": (4 * arr, "L1 L2 g/D par"),
    "k_mlpg_gv_gang": (1 * arr, "par"), "k_mc2b_mt": (B * T * L * 8.0, "bcoef [frame][35]"),
    "k_vocoder_lt": (samples * 8.0, "PCM f64"),
}
allk = {}
for k in sorted(acc["FETCH_SIZE"]):
    base = k.replace("jb::", "").split("<")[0]
    sh = shape(k)
    f = cal.get(sh) or cal.get(RD8) or 0.5
    raw = acc["FETCH_SIZE"][k]
    rec = {"fetch_bytes_raw": raw, "access_shape": sh, "reported_over_known": f, "fetch_bytes": raw / f,
           "write_bytes": acc["WRITE_SIZE"][k], "launches": cnt[k]}
    if base in must_read:
        rec["expected_read_bytes"], rec["expected_read_what"] = must_read[base]
        rec["implied_factor_if_compulsory_only"] = raw / must_read[base][0]
        rec["reads_ge_expected"] = bool(rec["fetch_bytes"] >= 0.98 * must_read[base][0])
    if base.startswith("k_vocoder"):
        rec["fetch_bytes_lower_bound"] = raw
        rec["fetch_note"] = ("lane-private 8-byte reads: if some of its requests are partial lines (64 B) they are tallied "
                             "in full and the true figure lies between the raw tally and its double")
    if base in must_write:
        rec["expected_write_bytes"], rec["expected_write_what"] = must_write[base]
    allk[k] = rec
dom = max((k for k in allk if "k_vocoder" in k), key=lambda k: allk[k]["write_bytes"] + allk[k]["fetch_bytes"])
tot_f = sum(v["fetch_bytes"] for v in allk.values()); tot_w = sum(v["write_bytes"] for v in allk.values())
res = {
    "batch": B, "frames": T, "kernel": dom,
    "kernel_sources_sha16": bench.get("kernel_sources_sha16"),
    "calibration_reported_over_known": cal,
    "whole_step_hbm_bytes": tot_f + tot_w,
    "whole_step_fetch_bytes": tot_f, "whole_step_fetch_bytes_raw": sum(acc["FETCH_SIZE"].values()),
    "whole_step_write_bytes": tot_w,
    "fetch_bytes_raw": allk[dom]["fetch_bytes_raw"], "fetch_bytes": allk[dom]["fetch_bytes"], "write_bytes": allk[dom]["write_bytes"],
    "hbm_bytes_per_launch": allk[dom]["fetch_bytes"] + allk[dom]["write_bytes"],
    "algorithmic_bytes_per_launch": 8.67 * samples,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB x 1024), bench.py --steps 1 --warmup 0. "
            "fetch_bytes = fetch_bytes_raw / reported_over_known, the factor measured in the same call on known byte counts "
            "(tools/microbench/fetch_calib.hip: every coalesced shape -- 4, 8, 16 B per lane, 128-byte pieces of far-apart rows "
            "-- reports 0.500 of its bytes); expected_read_bytes = the arrays a kernel must read (DESIGN.md section 3).  "
            "WRITE_SIZE is taken as reported (1.000 for 8 and 16 B per lane in the same file).",
    "all_kernels": allk,
}
json.dump(res, open(f"profiles/{R}_traffic.json", "w"), indent=1)
json.dump(res, open(f"{out}/{R}_traffic.json", "w"), indent=1)
print(f"dominant {dom}: fetch {res['fetch_bytes']/1e9:.2f} GB (raw {res['fetch_bytes_raw']/1e9:.2f}) write {res['write_bytes']/1e9:.2f} GB; "
      f"whole step fetch {tot_f/1e9:.1f} GB write {tot_w/1e9:.1f} GB = {(tot_f+tot_w)/1e9:.1f} GB")
for k, v in sorted(allk.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]))[:12]:
    e = v.get("expected_read_bytes")
    print(f"  {k[:34]:34s} read {v['fetch_bytes']/1e9:6.2f} GB (raw {v['fetch_bytes_raw']/1e9:5.2f} / {v['reported_over_known']:.2f})"
          + (f" must read {e/1e9:5.2f}" if e else " " * 16) + f"  write {v['write_bytes']/1e9:6.2f} GB")
PY
