"""Latency of Engine::synthesize for one short utterance (labels -> PCM on the host), warm."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import jbonsai_amd as J  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2  # noqa: E402

eng = J.Engine.load([VOICE])
for name, lab in (("bonsai (1.4 s)", SAMPLE_SENTENCE_1), ("is this bonsai (2.1 s)", SAMPLE_SENTENCE_2),
                  ("x10 (21 s)", list(SAMPLE_SENTENCE_2) * 10)):
    eng.synthesize(lab)
    ts = []
    for _ in range(20):
        t = time.perf_counter()
        out = eng.synthesize(lab)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    print(f"{name:24s}: {len(out) / 48000:.2f} s of audio, median {np.median(ts):.2f} ms, min {ts.min():.2f} ms, "
          f"{len(out) / 48000 / np.median(ts) * 1e3:.0f}x real time")

# streaming generator (src/speech.rs:65-82): time to the first 240 samples, then per step; the reference's CPU
# path needs ~31 us per frame (0.80 s per 25,546 frames on one i5-13500 core, README.md:84)
from tests.golden.labels import GENJI  # noqa: E402


def stream(labels, n_per_call):
    buf = np.zeros(240 * n_per_call)
    best = None
    for _ in range(4):
        t = time.perf_counter()
        g = eng.generator(labels)
        t_new = time.perf_counter() - t
        g.generate_step(buf)
        t_first = time.perf_counter() - t
        t = time.perf_counter()
        n = 1
        while True:
            r = g.generate_steps(buf, n_per_call) if n_per_call > 1 else g.generate_step(buf)
            if r <= 0:
                break
            n += r // 240
        t_rest = time.perf_counter() - t
        cur = (t_first + t_rest, t_new, t_first, t_rest, n)
        best = cur if best is None or cur[0] < best[0] else best
    _, t_new, t_first, t_rest, n = best
    return (f"generator() returns after {t_new * 1e3:.2f} ms, first frame after {t_first * 1e3:.2f} ms, then "
            f"{t_rest / (n - 1) * 1e6:.1f} us per 5 ms frame ({n} frames; whole utterance "
            f"{(t_first + t_rest) / n * 1e6:.1f} us per frame)")


for name, lab in (("bonsai (1.4 s)", SAMPLE_SENTENCE_1), ("x10 (21 s)", list(SAMPLE_SENTENCE_2) * 10), ("genji (164 s)", GENJI)):
    print(f"{name:16s} generate_step        : {stream(lab, 1)}")
    print(f"{name:16s} generate_steps(n = 8): {stream(lab, 8)}")

# eight generators in flight (made back to back, drained in turn)
t = time.perf_counter()
gs = [eng.generator(list(SAMPLE_SENTENCE_2) * 10) for _ in range(8)]
buf = np.zeros(240 * 8)
n = 0
for g in gs:
    while True:
        r = g.generate_steps(buf, 8)
        if r <= 0:
            break
        n += r // 240
dt = time.perf_counter() - t
print(f"8 generators x 21 s: {dt * 1e3:.1f} ms for {n} frames = {dt / n * 1e6:.2f} us per frame")
