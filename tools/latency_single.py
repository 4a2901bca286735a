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

# streaming generator (src/speech.rs:65-82): time to the first 240 samples, then per step
buf = np.zeros(240)
for _ in range(3):
    t = time.perf_counter()
    g = eng.generator(SAMPLE_SENTENCE_1)
    g.generate_step(buf)
    t_first = time.perf_counter() - t
    t = time.perf_counter()
    n = 0
    while g.generate_step(buf) > 0:
        n += 1
    t_rest = time.perf_counter() - t
print(f"generator: first frame after {t_first * 1e3:.2f} ms, then {t_rest / n * 1e6:.0f} us per 5 ms frame ({n} frames)")
