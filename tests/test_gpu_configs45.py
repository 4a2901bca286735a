"""BASELINE configs 4 and 5 at their stated batch (1024), with two DIFFERENT voices for config 5, and the
reference tree's long real label sequences, on the GPU through the C ABI.

  config 4 ("tohoku-f01, mcep+lf0+bap, batch=1024"; the voice is absent from the reference tree: the
      documented substitute is nitech with all three streams) -- label level: 1024 distinct
      label utterances through jb_synthesize_batch; state level: 1024 distinct synthetic utterances
      created from pdf row indices.
  config 5 ("two-voice interpolation, batch=1024") -- [nitech, permuted nitech]
      (tests/golden/make_permuted_voice.py), weights as in the reference's two-voice test
      (src/lib.rs:80-84: 0.7/0.3 duration, MCP, LF0; 1.0/0.0 LPF) and 0.5/0.5; the device gather + blend
      (k_gather_blend) against the oracle's VoiceSet::weighted (voice_set.rs:80-95) on >= 8 utterances of
      the batch, and bitwise against the host blend (JB_HOST_BLEND=1).

PARITY UNPINNED for the blend of two different voices (the reference's `bonsai_multi` golden needs the
absent tohoku-f01 files); everything single-voice here is pinned through the oracle's goldens."""
import os

import numpy as np
import pytest

import jbonsai_amd as J
from jbonsai_amd import synth
from oracle import oracle as O
from tests.conftest import VOICE
from tests.golden.labels import BENCH_LETTER, GENJI, label_pool_utterances
from tests.golden.make_permuted_voice import permuted_voice_path
from tests.helpers import rel_rms, PCM_TOL
from tests.test_gpu_configs import oracle_pcm

pytestmark = pytest.mark.gpu
B = 1024
W_REF = {"duration": [0.7, 0.3], "parameter": [[0.7, 0.3], [0.7, 0.3], [1.0, 0.0]],
         "gv": [[0.7, 0.3], [0.7, 0.3], [1.0, 0.0]]}
W_HALF = {"duration": [0.5, 0.5], "parameter": [[0.5, 0.5]] * 3, "gv": [[0.5, 0.5]] * 3}
PICKS = (0, 1, 100, 255, 256, 511, 777, 1000, 1023)


@pytest.fixture(scope="module")
def voice2(tmp_path_factory):
    return permuted_voice_path(tmp_path_factory.mktemp("voice2"))


@pytest.fixture(scope="module")
def pool():
    return label_pool_utterances(B)


def set_weights(e, w):
    e.condition.set_interpolation_duration(w["duration"])
    for s in range(3):
        e.condition.set_interpolation_parameter(s, w["parameter"][s])
        if s < 2:
            e.condition.set_interpolation_gv(s, w["gv"][s])


def with_env(name, value, fn):
    os.environ[name] = value
    try:
        return fn()
    finally:
        del os.environ[name]


# ---- label level -----------------------------------------------------------------------------------
def test_config4_labels_batch_1024(pool, oracle_voice):
    """1024 distinct label utterances (seeded spans of the genji label pool), single nitech voice with
    its three streams, one jb_synthesize_batch call; nine of them against the oracle."""
    e = J.Engine.load([VOICE])
    outs = e.synthesize_batch(pool)
    assert len(outs) == B and all(len(o) % 240 == 0 and len(o) > 0 for o in outs)
    for i in PICKS:
        ref = oracle_voice.synthesize(pool[i])
        assert len(outs[i]) == len(ref)
        assert rel_rms(outs[i], ref) <= PCM_TOL, i
    total = sum(len(o) for o in outs)
    print("config 4 labels: 1024 utterances,", total / 48000.0, "s of audio")


@pytest.mark.parametrize("weights", [W_REF, W_HALF], ids=["0.7/0.3+1/0", "0.5/0.5"])
def test_config5_labels_batch_1024_two_different_voices(pool, voice2, weights):
    e = J.Engine.load([VOICE, voice2])
    set_weights(e, weights)
    vs = O.VoiceSet([VOICE, voice2], weights)
    outs = e.synthesize_batch(pool)                               # device gather + blend
    host = with_env("JB_HOST_BLEND", "1", lambda: e.synthesize_batch(pool))
    assert len(outs) == B
    for a, b in zip(outs, host):
        assert np.array_equal(a, b)
    for i in PICKS:
        ref = vs.synthesize(pool[i])
        assert len(outs[i]) == len(ref), i
        assert rel_rms(outs[i], ref) <= PCM_TOL, i
    # the blend is a real one: neither voice alone gives this audio, nor does the swapped order
    one = O.Voice(VOICE).synthesize(pool[0])
    assert len(one) != len(outs[0]) or rel_rms(outs[0], one) > 1e-3
    if weights is W_REF:
        swapped = O.VoiceSet([voice2, VOICE], weights).synthesize(pool[0])
        assert len(swapped) != len(outs[0]) or rel_rms(outs[0], swapped) > 1e-3


def test_two_voice_single_utterance_entries(voice2):
    """jb_synthesize and the streaming generator over two different voices (host blend path)."""
    e = J.Engine.load([VOICE, voice2])
    set_weights(e, W_REF)
    e.condition.set_speed(1.2)
    e.condition.set_additional_half_tone(1.5)
    vs = O.VoiceSet([VOICE, voice2], W_REF)
    ref = vs.synthesize(BENCH_LETTER, speed=1.2, half_tone=1.5)
    got = e.synthesize(BENCH_LETTER)
    assert len(got) == len(ref) and rel_rms(got, ref) <= PCM_TOL
    g = e.generator(BENCH_LETTER)
    assert rel_rms(g.generate_all(), ref) <= PCM_TOL


# ---- state level (what bench.py's config4 / config5 records run) ------------------------------------
def _lens_1024():
    return synth.mixed_lengths(B, lo=200, hi=3000, seed=45)


def test_config4_state_level_batch_1024():
    eng = J.Engine.load([VOICE])
    tab, vi = synth.VoiceTables(eng), eng.voice_info()
    lens = _lens_1024()
    pset = tab.pdf_set()
    utts = [synth.synth_utterance(tab, T, 4000 + i, indexed=True) for i, T in enumerate(lens)]
    with J.Batch(vi, utts, pdf_set=pset) as b:
        b.run()
        b.sync()
        info = b.info()
        got = {i: b.pcm(i) for i in PICKS}
    print("config 4 state level:", info, sum(lens), "frames")
    for i in PICKS:
        ref, _ = oracle_pcm(vi, synth.synth_utterance(tab, lens[i], 4000 + i))
        assert len(got[i]) == lens[i] * 240 and rel_rms(got[i], ref) <= PCM_TOL, i
    pset.close()


@pytest.mark.parametrize("weights", [W_REF, W_HALF], ids=["0.7/0.3+1/0", "0.5/0.5"])
def test_config5_state_level_batch_1024_two_different_voices(voice2, weights):
    """Indexed batch over a two-voice pdf set: rows of BOTH tables gathered and blended on the device;
    the oracle is given the blend formed on the host in the reference's order."""
    eng = J.Engine.load([VOICE, voice2])
    tabs = [synth.VoiceTables(eng, 0), synth.VoiceTables(eng, 1)]
    assert not np.array_equal(tabs[0].concat[0], tabs[1].concat[0])
    vi = eng.voice_info()
    lens = _lens_1024()
    pset = synth.voice_set_pdf_set(tabs)
    utts = [synth.synth_utterance_voices(tabs, weights, T, 5000 + i, indexed=True) for i, T in enumerate(lens)]
    with J.Batch(vi, utts, pdf_set=pset, keep_tracks=True) as b:
        b.run()
        b.sync()
        got = {i: b.pcm(i) for i in PICKS}
        trk = {i: [b.track(i, s) for s in range(3)] for i in PICKS[:3]}
    for i in PICKS:
        u = synth.synth_utterance_voices(tabs, weights, lens[i], 5000 + i)
        ref, tr = oracle_pcm(vi, u)
        assert len(got[i]) == lens[i] * 240 and rel_rms(got[i], ref) <= PCM_TOL, i
        if i in trk:
            for s in range(3):
                np.testing.assert_allclose(trk[i][s], tr[s], rtol=1e-12, atol=1e-13)
    # a swapped voice order must NOT pass (the test can see a voice-index mix-up)
    if weights is W_REF:
        u_sw = synth.synth_utterance_voices(tabs[::-1], weights, lens[PICKS[0]], 5000 + PICKS[0])
        ref_sw, _ = oracle_pcm(vi, u_sw)
        assert len(ref_sw) != len(got[PICKS[0]]) or rel_rms(got[PICKS[0]], ref_sw) > 1e-3
    pset.close()


# ---- the reference tree's long real label sequences -------------------------------------------------
def test_genji_1456_labels(oracle_voice):
    """examples/genji/genji.lab (1,456 labels, 32,865 frames = 164 s with the nitech voice) through
    jb_synthesize, and inside a batch next to the 43-label sentence of benches/bonsais.rs."""
    e = J.Engine.load([VOICE])
    ref = oracle_voice.synthesize(GENJI)
    assert len(ref) == 32865 * 240
    got = e.synthesize(GENJI)
    assert len(got) == len(ref) and rel_rms(got, ref) <= PCM_TOL
    ref43 = oracle_voice.synthesize(BENCH_LETTER)
    outs = e.synthesize_batch([BENCH_LETTER, GENJI, [], GENJI[:700], BENCH_LETTER])
    assert [len(o) for o in outs][:3] == [len(ref43), len(ref), 0]
    assert rel_rms(outs[0], ref43) <= PCM_TOL and rel_rms(outs[1], ref) <= PCM_TOL
    assert np.array_equal(outs[0], outs[4])
    assert rel_rms(outs[3], oracle_voice.synthesize(GENJI[:700])) <= PCM_TOL


def test_bench_letter_43_labels(oracle_voice):
    e = J.Engine.load([VOICE])
    ref = oracle_voice.synthesize(BENCH_LETTER)
    got = e.synthesize(BENCH_LETTER)
    assert len(got) == len(ref) == 742 * 240 and rel_rms(got, ref) <= PCM_TOL
