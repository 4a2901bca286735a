"""Round 5 boundary additions (VERDICT r4 "next"):

  jb_batch_kernel_info            which vocoder kernel a batch's work list was built for (what bench.py names)
  IndexUtterance edited between two batches: the second batch sees the new arrays (the marshalled struct is cached)
  jb_release_cached_memory        also frees the pinned staging chunks of the upload arenas; batches work afterwards
"""
import numpy as np
import pytest

import jbonsai_amd as J
from jbonsai_amd import synth
from tests.conftest import VOICE
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import rel_rms, VERIFY_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert J.lib().jb_device_count() > 0
    eng = J.Engine.load([VOICE])
    return eng, synth.VoiceTables(eng), eng.voice_info()


def test_kernel_info_names_the_kernel_the_work_list_was_built_for(ctx):
    eng, tab, vi = ctx
    small = [synth.synth_utterance(tab, 300, 11)]
    with J.Batch(vi, small) as b:
        b.run()
        b.sync()
        assert b.kernel_info() == ("k_vocoder", 0)  # one chunk per wave below 100 k frames per batch
    big = [synth.synth_utterance(tab, 2000, 20 + i) for i in range(4)] * 16  # 128 k frames
    with J.Batch(vi, big) as b:
        b.run()
        b.sync()
        name, waves = b.kernel_info()
        assert name == "k_vocoder_lt" and waves in (1, 2)
        assert waves == 1  # too few frames to give every SIMD two waves of chunks long against their warm-up
    with J.Batch(vi, big, kernel="wave") as b:
        b.run()
        b.sync()
        assert b.kernel_info() == ("k_vocoder", 0)


def test_index_utterance_edited_between_batches(ctx):
    eng, tab, vi = ctx
    pset = tab.pdf_set(0)
    try:
        u = synth.synth_utterance(tab, 400, 31, indexed=True)
        v = synth.synth_utterance(tab, 400, 31)  # the same utterance from state-level arrays
        with J.Batch(vi, [u], pdf_set=pset) as b, J.Batch(vi, [v]) as c:
            b.run(), c.run()
            b.sync(), c.sync()
            first = b.pcm(0)
            assert rel_rms(first, c.pcm(0)) <= 1e-12
        # halve every duration: the struct the first batch was made from must not be what the second one uploads
        u.durations = np.maximum(u.durations // 2, 1).astype(np.uint32)
        with J.Batch(vi, [u], pdf_set=pset) as b:
            b.run()
            b.sync()
            assert b.num_frames(0) == int(u.durations.sum()) < 400
            assert b.num_samples(0) == b.num_frames(0) * 240
    finally:
        pset.close()


def test_release_cached_memory_frees_the_pinned_chunks_and_batches_go_on(ctx):
    eng, tab, vi = ctx
    utts = [synth.synth_utterance(tab, 250 + 10 * i, 40 + i) for i in range(6)]
    with J.Batch(vi, utts) as b:
        b.run()
        b.sync()
        ref = [b.pcm(i) for i in range(len(utts))]
    assert J.lib().jb_release_cached_memory() == 0
    assert J.lib().jb_release_cached_memory() == 0  # (nothing left: still fine)
    with J.Batch(vi, utts) as b:
        b.run()
        b.sync()
        for i, r in enumerate(ref):
            assert np.array_equal(b.pcm(i), r)


def test_two_wave_vocoder_equals_the_wave_kernel_bit_for_bit(monkeypatch):
    """Launches of at most four items per CU run k_vocoder_pair (a producer wave for gain + df1 and the PCM stores, a
    consumer wave for df2, a block apart through LDS; up to two items per CU every wave has a SIMD of its own, up to
    four an item's two waves share one); JB_NO_PAIR_KERNEL=1 keeps k_vocoder.  Same operations in the
    same order: the PCM must be the same BITS -- single sentences through the engine, a ragged batch chunked (with
    its redo round), serial, with a 2-frame warm-up (dozens of chunks redone from saved states) and through the
    16-bit sink; the streaming generator's serially served frames too."""
    eng = J.Engine.load([VOICE])
    tab = synth.VoiceTables(eng)
    vi = eng.voice_info()

    def both(fn):
        monkeypatch.delenv("JB_NO_PAIR_KERNEL", raising=False)
        a = fn()
        monkeypatch.setenv("JB_NO_PAIR_KERNEL", "1")
        b = fn()
        monkeypatch.delenv("JB_NO_PAIR_KERNEL", raising=False)
        return a, b

    for lab in (SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2):
        a, b = both(lambda: eng.synthesize(lab))
        assert len(a) > 0 and np.array_equal(a, b)
    utts = [synth.synth_utterance(tab, T, 40 + T) for T in (300, 1, 77, 512, 150)]
    for kw in (dict(), dict(serial=True), dict(chunk_frames=24, warmup_frames=2, verify_tol=VERIFY_TOL), dict(pcm_i16=True),
               dict(chunk_frames=16, kernel="wave")):
        def run():
            with J.Batch(vi, utts, **kw) as bt:
                bt.run()
                bt.sync()
                return [bt.pcm_i16(i) if kw.get("pcm_i16") else bt.pcm(i) for i in range(len(utts))], bt.info()
        (a, ia), (b, ib) = both(run)
        assert ia["n_redo"] == ib["n_redo"]
        for x, y in zip(a, b):
            assert np.array_equal(x, y), kw

    # 600-1000 chunks: the eight-wave form (four items per workgroup, an item's consumer and producer on one SIMD)
    rng = np.random.default_rng(3)
    utts8 = [synth.synth_utterance(tab, int(T), 900 + k) for k, T in enumerate(rng.integers(7, 1500, 36))]

    def run8():
        with J.Batch(vi, utts8) as bt:
            bt.run()
            bt.sync()
            return [bt.pcm(i) for i in range(len(utts8))], bt.info()
    (a, ia), (b, ib) = both(run8)
    assert 512 < ia["n_items"] <= 1024 and ia["n_redo"] == ib["n_redo"]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)

    def stream():
        g = eng.generator(SAMPLE_SENTENCE_1)
        fp = g.fperiod()
        out = np.zeros(g.total_frames() * fp)
        k = 0
        while True:
            n = g.generate_step(out[k:])
            if n == 0:
                break
            k += n
        return out[:k]
    a, b = both(stream)
    assert len(a) == 66480 and np.array_equal(a, b)
