"""Leak soak on the GPU box: thousands of requests of changing kind and size through the engine and the batch interface,
host RSS and free device memory sampled along the way.  The library pools device blocks, pinned staging chunks and
streams between requests (bounded: jb_set_cached_memory_limit, 16 pinned chunks); what must not happen is growth that
does not stop.

    python tools/leak_soak.py [--rounds 40] [--per-round 60] > profiles/rNN_leak_soak.txt
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import psutil

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2  # noqa: E402


def dev_free_mb(hip):
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return free.value / 2**20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--per-round", type=int, default=60)
    a = ap.parse_args()
    hip = ctypes.CDLL("libamdhip64.so")
    proc = psutil.Process()
    eng = J.Engine.load([VOICE])
    tab = synth.VoiceTables(eng)
    vi = eng.voice_info()
    rng = np.random.default_rng(99)
    sentences = [SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2, list(SAMPLE_SENTENCE_2) * 4, list(SAMPLE_SENTENCE_1) * 9]
    utt_pool = [synth.synth_utterance(tab, int(T), 300 + k) for k, T in enumerate(rng.integers(1, 3000, 24))]
    calls = 0
    t0 = time.perf_counter()
    rows, released = [], []
    for r in range(a.rounds):
        for _ in range(a.per_round):
            kind = int(rng.integers(0, 7))
            if kind == 0:
                eng.synthesize(sentences[int(rng.integers(0, len(sentences)))])
            elif kind == 1:
                eng.synthesize_batch([sentences[int(i)] for i in rng.integers(0, len(sentences), int(rng.integers(1, 9)))],
                                     i16=bool(rng.integers(0, 2)))
            elif kind == 2:
                us = [utt_pool[int(i)] for i in rng.integers(0, len(utt_pool), int(rng.integers(1, 12)))]
                with J.Batch(vi, us, pcm_i16=bool(rng.integers(0, 2))) as b:
                    b.run()
                    b.sync()
            elif kind == 3:
                us = [utt_pool[int(i)] for i in rng.integers(0, len(utt_pool), 3)]
                with J.Batch(vi, us, chunk_frames=24, warmup_frames=2, keep_tracks=True) as b:  # redo-heavy, debug taps
                    b.run()
                    b.sync()
                    b.excitation(0)
            elif kind == 4:
                g = eng.generator(sentences[int(rng.integers(0, 2))])
                buf = np.zeros(g.fperiod() * 8)
                for _k in range(int(rng.integers(1, 40))):
                    if g.generate_steps(buf, 8) == 0:
                        break
                del g
            elif kind == 5:
                e2 = eng.clone()
                e2.synthesize(SAMPLE_SENTENCE_1)
                e2.close()
            else:
                try:
                    eng.synthesize(["not a label"])
                except J.JbError:
                    pass
            calls += 1
        rows.append((calls, proc.memory_info().rss / 2**20, dev_free_mb(hip)))
        print(f"after {calls:5d} requests: host RSS {rows[-1][1]:8.1f} MB, device free {rows[-1][2]:10.1f} MB", flush=True)
        if r + 1 == a.rounds // 2 or r + 1 == a.rounds:
            # what the pools hold is not a leak: empty them at half time and at the end and compare what is left in use
            J.lib().jb_release_cached_memory()
            released.append((proc.memory_info().rss / 2**20, dev_free_mb(hip)))
            print(f"      jb_release_cached_memory: host RSS {released[-1][0]:8.1f} MB, device free {released[-1][1]:10.1f} MB", flush=True)
    half = len(rows) // 2
    rss_growth = rows[-1][1] - rows[half][1]
    dev_leak = released[0][1] - released[1][1]
    print(f"{calls} requests in {time.perf_counter() - t0:.1f} s; host RSS over the second half {rss_growth:+.1f} MB; device memory "
          f"still in use after the pools were emptied, end against half time: {dev_leak:+.1f} MB")
    ok = rss_growth < 64 and abs(dev_leak) < 16
    print("NO GROWTH" if ok else "GROWTH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
