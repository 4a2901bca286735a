"""Engine-level end to end (labels -> PCM on the host) for a batch of long utterances:
jb_synthesize_batch = host front half on worker threads (label parse, tree search, durations) +
device gather/blend + the GPU hot path + D2H of the f64 PCM.  Prints wall times; JB_HOST_BLEND=1
and JB_HOST_THREADS=n select the alternatives."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_2  # noqa: E402

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 75  # 75 x 20 labels ~ 128 s
eng = J.Engine.load([VOICE])
utt = list(SAMPLE_SENTENCE_2) * reps
batch = [utt] * n_utts
eng.synthesize_batch(batch[:2])  # warm-up: device tables, noise table
for label, env in (("device gather, threads", {}), ("same, 16-bit sink", {"I16": "1"}),
                   ("host blend, threads", {"JB_HOST_BLEND": "1"}),
                   ("device gather, 1 thread", {"JB_HOST_THREADS": "1"})):
    os.environ.update(env)
    outs = None  # release the previous result outside the timed region
    t = time.perf_counter()
    outs = eng.synthesize_batch(batch, i16="I16" in env)
    dt = time.perf_counter() - t
    for k in env:
        del os.environ[k]
    ns = sum(len(o) for o in outs)
    print(f"{label:28s}: {n_utts} x {len(outs[0]) / 48000:.1f} s in {dt:.2f} s wall = {ns / dt / 1e6:.1f} Msamples/s "
          f"= {ns / dt / 48000:.0f}x real time")
