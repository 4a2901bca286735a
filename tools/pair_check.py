"""k_vocoder_pair (two waves per item) against k_vocoder (JB_NO_PAIR_KERNEL=1): bits of the PCM of single sentences,
of a small ragged batch through the Batch interface (chunked, serial, 16-bit) and of a streaming generator, and the
time of a single-sentence synthesis either way."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2  # noqa: E402

eng = J.Engine.load([VOICE])
tab = synth.VoiceTables(eng)
vi = eng.voice_info()


def both(fn):
    os.environ.pop("JB_NO_PAIR_KERNEL", None)
    a = fn()
    os.environ["JB_NO_PAIR_KERNEL"] = "1"
    b = fn()
    os.environ.pop("JB_NO_PAIR_KERNEL", None)
    return a, b


ok = True
for name, lab in (("sentence 1", SAMPLE_SENTENCE_1), ("sentence 2", SAMPLE_SENTENCE_2)):
    a, b = both(lambda: eng.synthesize(lab))
    same = np.array_equal(a, b)
    ok &= same
    print(f"{name}: {len(a)} samples, bits {'identical' if same else 'DIFFER'}")
utts = [synth.synth_utterance(tab, T, 40 + T) for T in (300, 1, 77, 512, 150)]
for kw in (dict(), dict(serial=True), dict(chunk_frames=24, warmup_frames=2, verify_tol=1e-9), dict(pcm_i16=True)):
    def run():
        with J.Batch(vi, utts, **kw) as bt:
            bt.run()
            bt.sync()
            return [bt.pcm_i16(i) if kw.get("pcm_i16") else bt.pcm(i) for i in range(len(utts))], bt.info()
    (a, ia), (b, ib) = both(run)
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    ok &= same
    print(f"batch {kw}: bits {'identical' if same else 'DIFFER'}; redo {ia['n_redo']} / {ib['n_redo']}")
# 600-1000 items: the eight-wave form (consumer and producer of an item on one SIMD)
rng = np.random.default_rng(3)
utts8 = [synth.synth_utterance(tab, int(T), 900 + k) for k, T in enumerate(rng.integers(7, 1500, 36))]
for kw in (dict(), dict(chunk_frames=24, warmup_frames=3, verify_tol=1e-9)):
    def run8():
        with J.Batch(vi, utts8, **kw) as bt:
            bt.run()
            bt.sync()
            return [bt.pcm(i) for i in range(len(utts8))], bt.info()
    (a, ia), (b, ib) = both(run8)
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    ok &= same
    print(f"36 utterances {kw}: {ia['n_items']} items, bits {'identical' if same else 'DIFFER'}; redo {ia['n_redo']} / {ib['n_redo']}")
    for k, env in enumerate((None, "8", "1")):
        os.environ.pop("JB_NO_PAIR_KERNEL", None)
        if env:
            os.environ["JB_NO_PAIR_KERNEL"] = env
        with J.Batch(vi, utts8, **kw) as bt:
            ts = []
            for _ in range(12):
                t0 = time.perf_counter()
                bt.run()
                bt.sync()
                ts.append(time.perf_counter() - t0)
        print(f"   {('eight-wave pair', 'k_vocoder (JB_NO_PAIR_KERNEL=8)', 'k_vocoder (JB_NO_PAIR_KERNEL=1)')[k]}: run+sync median {np.median(ts[2:]) * 1e3:.3f} ms")
    os.environ.pop("JB_NO_PAIR_KERNEL", None)
for name, lab in (("sentence 1", SAMPLE_SENTENCE_1), ("sentence 2 x 5", list(SAMPLE_SENTENCE_2) * 5)):
    for k in range(2):
        if k:
            os.environ["JB_NO_PAIR_KERNEL"] = "1"
        for _ in range(5):
            eng.synthesize(lab)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            eng.synthesize(lab)
            ts.append(time.perf_counter() - t0)
        print(f"{name}: {'k_vocoder     ' if k else 'k_vocoder_pair'} median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms")
        os.environ.pop("JB_NO_PAIR_KERNEL", None)
print("ALL IDENTICAL" if ok else "DIFFERENCES")
sys.exit(0 if ok else 1)
