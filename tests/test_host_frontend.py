"""Host logic of the product (C++ front half behind jb_engine_*): voice parsing,
tree search, pdf gather, durations, label alignment, Condition clamps.  No GPU.

Checked (a) against the reference's own pins (src/model/mod.rs:183-392,
src/duration.rs:144-179) and (b) bit-for-bit against the oracle's independent
restatement on the same labels."""
import numpy as np
import pytest

import jbonsai_amd as J
from oracle import oracle as O
from tests.conftest import VOICE
from tests.golden.labels import ALIGNED_1, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.test_oracle_golden import DUR_10, DUR_12, DUR_AL


@pytest.fixture(scope="module")
def engine():
    return J.Engine.load([VOICE])


def test_metadata(engine):
    c = engine.condition
    assert (c.get_sampling_frequency(), c.get_fperiod()) == (48000, 240)
    assert (engine.num_voices, engine.num_streams, engine.num_states) == (1, 3, 5)
    assert c.get_alpha() == 0.55 and c.get_beta() == 0.0 and c.get_speed() == 1.0
    assert [c.get_msd_threshold(i) for i in range(3)] == [0.5] * 3
    assert [c.get_gv_weight(i) for i in range(3)] == [1.0] * 3
    vi = engine.voice_info()
    assert [s.vector_length for s in vi.streams] == [35, 1, 31]
    assert vi.streams[0].windows == [[1.0], [-0.5, 0.0, 0.5], [1.0, -2.0, 1.0]]


def test_load_from_bytes(engine):
    e2 = J.Engine.load_from_bytes([VOICE.read_bytes()])
    assert e2.tree_index(0, 2, SAMPLE_SENTENCE_1[2]) == (2, 144)


def test_load_errors():
    with pytest.raises(J.JbError) as ei:
        J.Engine.load([])
    assert ei.value.code == -4  # ModelError::EmptyVoice
    with pytest.raises(J.JbError) as ei:
        J.Engine.load(["/nonexistent.htsvoice"])
    assert ei.value.code == -4  # ModelError::Io
    with pytest.raises(J.JbError):
        J.Engine.load_from_bytes([b"[GLOBAL]\nnot a voice"])


def test_tree_index(engine):
    lab = SAMPLE_SENTENCE_1[2]
    assert engine.tree_index(0, 2, lab) == (2, 144)
    assert engine.tree_index(2, 2, lab) == (2, 234)
    assert engine.tree_index(5, 2, lab) == (2, 3)


def test_pdf_tables_equal_oracle(engine, oracle_voice):
    for kind in (0, 1, 2, 3, 4, 5):
        nt, _ = engine.model_shape(kind)
        assert nt == oracle_voice.ntree(kind)
        for t in range(nt):
            assert np.array_equal(engine.pdf_table(kind, t), oracle_voice.pdf_table(kind, t))


@pytest.mark.parametrize("labels", [SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2])
def test_states_equal_oracle(engine, oracle_voice, labels):
    u = engine.states(labels)
    assert u.durations.tolist() == oracle_voice.durations(labels).tolist()
    for i in range(3):
        o = oracle_voice.stream_states(i, labels)
        s = u.streams[i]
        assert np.array_equal(s.mean, o.mean) and np.array_equal(s.var, o.var)
        if o.is_msd:
            assert np.array_equal(s.msd, o.msd)
        else:
            assert s.msd is None
        if o.use_gv:
            assert np.array_equal(s.gv_mean, o.gv_mean) and np.array_equal(s.gv_var, o.gv_var)
            assert np.array_equal(s.gv_switch, o.gv_switch)
        else:
            assert s.gv_mean is None


def test_durations_reference_pins(engine):
    e = J.Engine.load([VOICE])
    assert e.states(SAMPLE_SENTENCE_1).durations.tolist() == DUR_10
    e.condition.set_speed(1.2)
    assert e.states(SAMPLE_SENTENCE_1).durations.tolist() == DUR_12
    e.condition.set_speed(1.0)
    e.condition.set_phoneme_alignment_flag(True)
    assert e.states(ALIGNED_1).durations.tolist() == DUR_AL
    e.condition.set_phoneme_alignment_flag(False)
    assert e.states(ALIGNED_1).durations.tolist() == DUR_10  # times ignored


def test_speed_14_matches_oracle(oracle_voice):
    e = J.Engine.load([VOICE])
    e.condition.set_speed(1.4)
    d = e.states(SAMPLE_SENTENCE_2).durations
    assert d.tolist() == oracle_voice.durations(SAMPLE_SENTENCE_2, 1.4).tolist()
    assert int(d.sum()) * 240 == 72000  # src/lib.rs:141


def test_half_tone_and_condition_clamps():
    e = J.Engine.load([VOICE])
    c = e.condition
    base = e.states(SAMPLE_SENTENCE_1).streams[1].mean.copy()
    c.set_additional_half_tone(2.0)
    m = e.states(SAMPLE_SENTENCE_1).streams[1].mean
    want = np.clip(base[:, 0] + 2.0 * 0.05776226504666211, 2.995732273553991, 9.903487552536127)
    assert np.array_equal(m[:, 0], want) and np.array_equal(m[:, 1:], base[:, 1:])
    c.set_msd_threshold(1, 7.0)
    c.set_gv_weight(0, -3.0)
    c.set_speed(0.0)
    c.set_alpha(2.0)
    c.set_beta(-1.0)
    c.set_sampling_frequency(0)
    assert c.get_msd_threshold(1) == 1.0 and c.get_gv_weight(0) == 0.0
    assert c.get_speed() == 1.0e-6 and c.get_alpha() == 1.0 and c.get_beta() == 0.0
    assert c.get_sampling_frequency() == 1
    c.set_volume(6.0)
    assert abs(c.get_volume() - 6.0) < 1e-12


def test_interpolation_weight_errors():
    e = J.Engine.load([VOICE])
    e.condition.set_interpolation_duration([1.0])
    with pytest.raises(J.JbError) as ei:
        e.condition.set_interpolation_duration([0.7])
    assert ei.value.code == -7  # InvalidSum
    with pytest.raises(J.JbError) as ei:
        e.condition.set_interpolation_parameter(0, [0.5, 0.5])
    assert ei.value.code == -7  # InvalidLength


def test_interpolation_weight_getters_and_engine_new():
    """InterporationWeight::{get_duration, get_parameter, get_gv} (interporation_weight.rs:115-125) and
    Engine::new(voices, condition) / Engine::clone (engine.rs:246,289-291)."""
    e = J.Engine.load([VOICE, VOICE])
    c = e.condition
    assert c.get_interpolation_duration().tolist() == [0.5, 0.5]
    c.set_interpolation_duration([0.7, 0.3])
    c.set_interpolation_parameter(1, [0.25, 0.75])
    c.set_interpolation_gv(0, [1.0, 0.0])
    assert c.get_interpolation_duration().tolist() == [0.7, 0.3]
    assert c.get_interpolation_parameter(1).tolist() == [0.25, 0.75]
    assert c.get_interpolation_parameter(0).tolist() == [0.5, 0.5]
    assert c.get_interpolation_gv(0).tolist() == [1.0, 0.0]
    with pytest.raises(J.JbError):
        c.get_interpolation_parameter(3)
    c.set_speed(1.3)
    k = e.clone()
    assert k.condition.get_speed() == 1.3 and k.condition.get_interpolation_duration().tolist() == [0.7, 0.3]
    k.condition.set_speed(0.9)          # a copy of the condition, not a view
    assert c.get_speed() == 1.3
    assert k.states(SAMPLE_SENTENCE_1).durations.sum() > e.states(SAMPLE_SENTENCE_1).durations.sum()
    e.close()                           # the voices are shared and outlive the first engine
    assert k.states(SAMPLE_SENTENCE_1).durations.size == 40
    one = J.Engine.load([VOICE])
    with pytest.raises(J.JbError) as ei:
        J.Engine.new(k, one)            # a condition made for one voice over a set of two
    assert ei.value.code == -7
    mixed = J.Engine.new(one, J.Engine.load([VOICE]))
    assert mixed.num_voices == 1


def test_two_voice_blend_matches_reference_formula(oracle_voice):
    """VoiceSet::weighted (voice_set.rs:80-95) with the same voice twice: the blend
    w0*p + w1*p must reproduce first*w0 then += w1*p."""
    e = J.Engine.load([VOICE, VOICE])
    e.condition.set_interpolation_duration([0.7, 0.3])
    for s in range(3):
        e.condition.set_interpolation_parameter(s, [0.7, 0.3])
    u = e.states(SAMPLE_SENTENCE_1)
    o = oracle_voice.stream_states(0, SAMPLE_SENTENCE_1)
    want = o.mean * 0.7
    want += 0.3 * o.mean
    assert np.array_equal(u.streams[0].mean, want)


def test_multiple_models():
    """src/model/mod.rs:395-428 through the product's host front half (jb_engine_states, jb_states_duration_params):
    the blended duration pdfs and the first LF0 state of two DIFFERENT voices, exact.  Needs no GPU.  Skipped until
    the tohoku-f01 files are supplied (tests/conftest.py: JB_TOHOKU_DIR)."""
    from tests.conftest import (MULTIPLE_MODELS_DURATION, MULTIPLE_MODELS_LF0_STATE0, MULTIPLE_MODELS_WEIGHTS,
                                tohoku_voices)

    e = J.Engine.load(tohoku_voices())
    e.condition.set_interpolation_duration(MULTIPLE_MODELS_WEIGHTS["duration"])
    e.condition.set_interpolation_parameter(1, MULTIPLE_MODELS_WEIGHTS["parameter"][1])
    lab = [SAMPLE_SENTENCE_1[2]]
    assert e.duration_params(lab).tolist() == [list(x) for x in MULTIPLE_MODELS_DURATION]
    st = e.states(lab).streams[1]
    (w0, w1, w2), msd = MULTIPLE_MODELS_LF0_STATE0
    for w, (m, v) in enumerate((w0, w1, w2)):
        assert st.mean[0, w] == m and st.var[0, w] == v
    assert st.msd[0] == msd


def test_duration_params_equal_oracle(engine, oracle_voice):
    """jb_states_duration_params = Models::duration(): the nitech pins of src/model/mod.rs:234-262 through the product,
    and the whole table against the oracle."""
    d = engine.duration_params(SAMPLE_SENTENCE_1)
    assert d.shape == (40, 2)
    assert d[0].tolist() == [7.939206123352051, 145.76211547851563]
    assert d[14].tolist() == [2.7264480590820313, 3.725647211074829]
    assert np.array_equal(d, oracle_voice.duration_params(SAMPLE_SENTENCE_1))


def test_label_errors(engine):
    with pytest.raises(J.JbError) as ei:
        engine.states(["0 100"])
    assert ei.value.code == -5
    with pytest.raises(J.JbError) as ei:
        engine.states(["abc def " + SAMPLE_SENTENCE_1[0]])
    assert ei.value.code == -5
    with pytest.raises(J.JbError) as ei:
        engine.states(["not-a-label"])
    assert ei.value.code == -5
    assert engine.states([]).durations.size == 0
    assert engine.states(["", SAMPLE_SENTENCE_1[0]]).durations.size == 5  # empty lines skipped


def test_synthesize_fails_loudly_without_gpu(engine):
    if J.lib().jb_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(J.JbError) as ei:
        engine.synthesize(SAMPLE_SENTENCE_1)
    assert ei.value.code == -3


def test_threaded_label_loops_do_not_change_the_states(monkeypatch):
    """A single long utterance splits its per-label work (tree searches, pdf blend) over host threads
    (JB_HOST_THREADS): the states must not depend on the thread count, and a model error in one piece
    must surface."""
    from tests.golden.labels import GENJI

    e = J.Engine.load([VOICE])
    outs = []
    for nt in ("1", "3", "8"):
        monkeypatch.setenv("JB_HOST_THREADS", nt)
        outs.append(e.states(GENJI))
    for o in outs[1:]:
        assert np.array_equal(o.durations, outs[0].durations)
        for a, b in zip(o.streams, outs[0].streams):
            assert np.array_equal(a.mean, b.mean) and np.array_equal(a.var, b.var)
            assert (a.msd is None) == (b.msd is None) and (a.msd is None or np.array_equal(a.msd, b.msd))
    bad = list(GENJI)
    bad[1000] = "not a label"
    with pytest.raises(J.JbError) as ei:
        e.states(bad)
    assert ei.value.code == -5
