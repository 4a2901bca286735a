"""Two DIFFERENT voices (BASELINE configs 4/5): the permuted second voice of
tests/golden/make_permuted_voice.py, the oracle's VoiceSet (voice_set.rs:80-95) and the product's host
front half against it, bit for bit.  No GPU.  PARITY UNPINNED for the blend of two different voices: the
reference's only golden for it (`bonsai_multi`, src/lib.rs:77-91) needs the absent tohoku-f01 files; the
oracle's blend is held by the single-voice goldens (it is the same code with nv = 1) and by the numpy
restatement of the formula below."""
import numpy as np
import pytest

import jbonsai_amd as J
from oracle import oracle as O
from tests.conftest import VOICE
from tests.golden.labels import BENCH_LETTER, GENJI, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.golden.make_permuted_voice import permuted_voice_path

W_REF = {"duration": [0.7, 0.3], "parameter": [[0.7, 0.3], [0.7, 0.3], [1.0, 0.0]],
         "gv": [[0.7, 0.3], [0.7, 0.3], [1.0, 0.0]]}  # src/lib.rs:80-84


@pytest.fixture(scope="module")
def voice2(tmp_path_factory):
    return permuted_voice_path(tmp_path_factory.mktemp("voice2"))


def set_weights(e, w):
    e.condition.set_interpolation_duration(w["duration"])
    for s in range(3):
        e.condition.set_interpolation_parameter(s, w["parameter"][s])
        if s < 2:
            e.condition.set_interpolation_gv(s, w["gv"][s])


def test_permuted_voice_differs_in_every_model(voice2):
    a, b = O.Voice(VOICE), O.Voice(voice2)
    assert (a.fs, a.fperiod, a.nstate, a.nstream, a.alpha, a.windows) == (b.fs, b.fperiod, b.nstate, b.nstream,
                                                                            b.alpha, b.windows)
    for kind in (0, 1, 2, 3, 4, 5):
        assert a.ntree(kind) == b.ntree(kind)
        for t in range(a.ntree(kind)):
            ta, tb = a.pdf_table(kind, t), b.pdf_table(kind, t)
            assert ta.shape == tb.shape and not np.array_equal(ta, tb)
            if len(ta) > 1:  # a permutation of the rows
                assert sorted(map(bytes, ta)) == sorted(map(bytes, tb))
    # the trees are untouched: every label reaches the same leaf number
    for lab in SAMPLE_SENTENCE_1[:4]:
        for kind in (0, 1, 2, 3):
            assert a.get_index(kind, 2, lab) == b.get_index(kind, 2, lab)
    # deterministic
    assert permuted_voice_path(voice2.parent).read_bytes() == voice2.read_bytes()


def test_oracle_voice_set_single_voice_is_the_pinned_path():
    one = O.Voice(VOICE).synthesize(SAMPLE_SENTENCE_1)
    assert np.array_equal(O.VoiceSet([VOICE]).synthesize(SAMPLE_SENTENCE_1), one)
    assert abs(one[30000] - -980.6757547598129) < 1e-10  # src/lib.rs:46


def test_oracle_blend_is_the_reference_formula(voice2):
    """first.mul(w0), then mul_add_assign(w1, second) -- plain multiply, then add (model.rs:111-129)."""
    a, b = O.Voice(VOICE), O.Voice(voice2)
    vs = O.VoiceSet([VOICE, voice2], W_REF)
    for i in range(3):
        sa, sb, sm = (x.stream_states(i, SAMPLE_SENTENCE_2) for x in (a, b, vs))
        w0, w1 = W_REF["parameter"][i]
        for name in ("mean", "var") + (("msd",) if sa.is_msd else ()):
            want = getattr(sa, name) * w0
            want += w1 * getattr(sb, name)
            assert np.array_equal(getattr(sm, name), want), (i, name)
        if not sa.is_msd:
            assert np.all(sm.msd == np.finfo(np.float64).max)
        if sa.use_gv:
            g0, g1 = W_REF["gv"][i]
            want = sa.gv_mean * g0
            want += g1 * sb.gv_mean
            assert np.array_equal(sm.gv_mean, want) and np.array_equal(sm.gv_switch, sa.gv_switch)
    da, db = a.duration_params(SAMPLE_SENTENCE_2), b.duration_params(SAMPLE_SENTENCE_2)
    mv = da * 0.7
    mv += 0.3 * db
    assert vs.durations(SAMPLE_SENTENCE_2).tolist() == [max(1, int(np.floor(m + 0.5))) for m in mv[:, 0]]
    # the voices really differ, and so do the two orders of an unequal blend
    assert not np.array_equal(a.stream_states(0, SAMPLE_SENTENCE_2).mean, b.stream_states(0, SAMPLE_SENTENCE_2).mean)
    rev = O.VoiceSet([voice2, VOICE], W_REF).stream_states(0, SAMPLE_SENTENCE_2)
    assert not np.array_equal(rev.mean, vs.stream_states(0, SAMPLE_SENTENCE_2).mean)


@pytest.mark.parametrize("weights", [W_REF, None], ids=["0.7/0.3+1/0", "0.5/0.5"])
@pytest.mark.parametrize("labels", [SAMPLE_SENTENCE_1, BENCH_LETTER, GENJI[:300]], ids=["s1", "letter", "genji300"])
def test_product_front_half_equals_oracle_voice_set(voice2, weights, labels):
    """jb_engine_states over [nitech, permuted]: durations and blended Gaussians bit for bit."""
    e = J.Engine.load([VOICE, voice2])
    assert e.num_voices == 2
    if weights is not None:
        set_weights(e, weights)
    vs = O.VoiceSet([VOICE, voice2], weights)
    u = e.states(labels)
    assert u.durations.tolist() == vs.durations(labels).tolist()
    for i in range(3):
        o, s = vs.stream_states(i, labels), u.streams[i]
        assert np.array_equal(s.mean, o.mean) and np.array_equal(s.var, o.var), i
        if o.is_msd:
            assert np.array_equal(s.msd, o.msd)
        if o.use_gv:
            assert np.array_equal(s.gv_mean, o.gv_mean) and np.array_equal(s.gv_var, o.gv_var)
            assert np.array_equal(s.gv_switch, o.gv_switch)


def test_product_front_half_two_voices_speed_and_half_tone(voice2):
    e = J.Engine.load([VOICE, voice2])
    set_weights(e, W_REF)
    e.condition.set_speed(1.3)
    vs = O.VoiceSet([VOICE, voice2], W_REF)
    assert e.states(BENCH_LETTER).durations.tolist() == vs.durations(BENCH_LETTER, 1.3).tolist()


def test_long_real_label_sequences_front_half(oracle_voice):
    """1,456 distinct real labels through the product's tree search, duration rule and pdf gather."""
    e = J.Engine.load([VOICE])
    for labels in (GENJI, BENCH_LETTER):
        u = e.states(labels)
        assert u.durations.tolist() == oracle_voice.durations(labels).tolist()
        for i in range(3):
            o = oracle_voice.stream_states(i, labels)
            assert np.array_equal(u.streams[i].mean, o.mean) and np.array_equal(u.streams[i].var, o.var)
    assert int(e.states(GENJI).durations.sum()) == 32865
