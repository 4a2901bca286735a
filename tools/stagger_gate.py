"""Gate of VERDICT r5 "next" 1 (hide parameter generation under the vocoder inside ONE step), with the
library as it is: BASELINE config 2 as G batches of 256/G copies on their own stream sets, the lane-triple
kernel at ONE wave per SIMD (four-wave workgroups, 256 VGPRs of every SIMD left free), started `delay` ms
apart; wall time of the G against one batch of 256.  Chunk geometry = the whole batch's (153 frames).

    python tools/stagger_gate.py [--groups 4] [--delays 0,2,5,8] [--reps 6] [--trace]

--trace: ONE staggered pass after the warm-up and nothing else (run under rocprofv3 --kernel-trace for a
timeline: tools/stagger_gate.sh)."""
import argparse
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--groups", type=int, default=4)
ap.add_argument("--delays", default="0,2,5,8")
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--chunk-frames", type=int, default=153)
ap.add_argument("--trace", action="store_true")
args = ap.parse_args()

import torch  # noqa: F401,E402  (one HIP runtime per process, as in bench.py)
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402

VOICE = ROOT / "tests" / "golden" / "voice" / "nitech_jp_atr503_m001.htsvoice"
eng = J.Engine.load([VOICE])
tab = synth.VoiceTables(eng)
vi = eng.voice_info()
utt = synth.synth_utterance(tab, synth.T_128S, 0)


def wall(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


def spin(ms):
    t1 = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t1:
        pass


G = args.groups
per = args.batch // G
groups = [J.Batch(vi, [utt] * per, kernel="triple", chunk_frames=args.chunk_frames) for _ in range(G)]
print("group batch:", per, "utterances,", groups[0].info(), groups[0].kernel_info(), flush=True)


def staggered(delay):
    def f():
        for i, b in enumerate(groups):
            if i and delay > 0:
                spin(delay)
            b.run()
        for b in groups:
            b.sync()
    return f


for b in groups:  # warm-up
    b.run()
    b.sync()
if args.trace:
    staggered(float(args.delays.split(",")[0]))()
    sys.exit(0)

whole = J.Batch(vi, [utt] * args.batch)
print("whole batch:", args.batch, "utterances,", whole.info(), whole.kernel_info(), flush=True)


def one():
    whole.run()
    whole.sync()


for _ in range(2):
    one()
med, lo = wall(one, args.reps)
print(f"one batch of {args.batch}: median {med:.2f} ms  min {lo:.2f} ms", flush=True)


def serial():
    for b in groups:
        b.run()
        b.sync()


med, lo = wall(serial, args.reps)
print(f"{G} groups one after the other: median {med:.2f} ms  min {lo:.2f} ms", flush=True)
for d in [float(x) for x in args.delays.split(",")]:
    med, lo = wall(staggered(d), args.reps)
    print(f"{G} groups, {d:4.1f} ms apart: median {med:.2f} ms  min {lo:.2f} ms   redo {[b.info()['n_redo'] for b in groups]}",
          flush=True)
med, lo = wall(one, args.reps)
print(f"one batch of {args.batch} (again): median {med:.2f} ms  min {lo:.2f} ms", flush=True)
