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
from tests.helpers import rel_rms

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
