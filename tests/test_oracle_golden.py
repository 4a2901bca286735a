"""Pins the CPU oracle (oracle/) against every golden vector the reference's
own tests hold for the hot path and its immediate inputs.

All numbers below are data from the reference's tests:
  src/lib.rs:39-160            PCM samples / lengths (abs eps 1e-10)
  src/duration.rs:144-179      duration vectors
  src/model/mod.rs:183-392     metadata, tree indices, duration pdfs, LF0 pdfs, LF0 GV
  src/mlpg_adjust/mask.rs:89-159  mask fill / boundary distances
  src/model/voice/window.rs:84-115  window widths
  src/label.rs:166-197         time alignment scaling
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import VOICE
from tests.golden.labels import ALIGNED_1, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2

EPS = 1.0e-10  # approx::assert_abs_diff_eq!(..., epsilon = 1.0e-10)


def test_metadata(oracle_voice):
    v = oracle_voice
    assert (v.fs, v.fperiod, v.nstate, v.nstream) == (48000, 240, 5, 3)
    assert v.alpha == 0.55 and v.stage == 0
    assert v.vector_length == [35, 1, 31]
    assert v.num_windows == [3, 3, 1]
    assert v.is_msd == [0, 1, 0] and v.use_gv == [1, 1, 0]
    assert v.windows[0] == [[1.0], [-0.5, 0.0, 0.5], [1.0, -2.0, 1.0]]
    assert v.windows[2] == [[1.0]]


def test_table_sizes(oracle_voice):
    # SURVEY Appendix A (measured from the file)
    v = oracle_voice
    assert [v.pdf_table(0, 0).shape] == [(245, 10)]
    assert [v.pdf_table(1, k).shape[0] for k in range(5)] == [183, 181, 162, 154, 210]
    assert [v.pdf_table(2, k).shape[0] for k in range(5)] == [371, 521, 451, 389, 420]
    assert [v.pdf_table(3, k).shape for k in range(5)] == [(1, 62)] * 5
    assert v.pdf_table(4, 0).shape == (2, 70) and v.pdf_table(5, 0).shape == (3, 2)


def test_tree_index(oracle_voice):
    v, lab = oracle_voice, SAMPLE_SENTENCE_1[2]
    assert v.get_index(0, 2, lab) == (2, 144)
    assert v.get_index(2, 2, lab) == (2, 234)
    assert v.get_index(5, 2, lab) == (2, 3)


def test_duration_pdfs(oracle_voice):
    d = oracle_voice.duration_params(SAMPLE_SENTENCE_1)
    assert d.shape == (40, 2)
    want = [
        (7.939206123352051, 145.76211547851563), (16.867250442504883, 353.91778564453125),
        (13.902158737182617, 178.05068969726563), (24.711565017700195, 395.954833984375),
        (15.016390800476074, 62.81060791015625), (2.9893455505371094, 3.7195587158203125),
        (3.650455951690674, 7.21462869644165), (2.317136287689209, 2.8865654468536377),
        (2.3675591945648193, 2.918273448944092), (2.4925434589385986, 2.9260120391845703),
        (2.1477856636047363, 2.4373505115509033), (3.2821402549743652, 4.192541599273682),
        (2.679042100906372, 3.923785924911499), (3.378859281539917, 3.866243362426758),
        (2.7264480590820313, 3.725647211074829),
    ]
    assert d[:15].tolist() == [list(x) for x in want]


def test_lf0_pdfs_and_gv(oracle_voice):
    st = oracle_voice.stream_states(1, SAMPLE_SENTENCE_1)
    assert st.mean.shape == (40, 3)
    want = [
        ((4.708907127380371, 0.027746843174099922), (0.010573429986834526, 0.0006717125070281327),
         (-0.019542237743735313, 0.002855533268302679), 0.05000000074505806),
        ((4.714630603790283, 0.03322882577776909), (-0.009544742293655872, 0.000757755886297673),
         (0.011145883239805698, 0.0031274918001145124), 0.05000000074505806),
        ((4.704207420349121, 0.040450580418109894), (0.004150974098592997, 0.0008980912389233708),
         (0.010611549019813538, 0.0024848130997270346), 0.05000000074505806),
        ((0.0, 1.0), (0.0, 1.0), (0.0, 1.0), 0.05000000074505806),
        ((4.768340110778809, 0.01530302595347166), (0.02272343635559082, 3.5269540603621863e-6),
         (-0.047215938568115234, 1.3166980352252722e-5), 0.05000000074505806),
        ((4.747085094451904, 0.009076375514268875), (-0.010534754022955894, 0.002568872645497322),
         (-0.016766104847192764, 0.014940978959202766), 0.23628035187721252),
        ((4.736148357391357, 0.009678148664534092), (0.00046353874495252967, 0.002193617168813944),
         (-0.01878436654806137, 0.013272966258227825), 0.3182770907878876),
        ((4.739607334136963, 0.0061369095928967), (0.014216499403119087, 0.001773378811776638),
         (0.014568353071808815, 0.008928200230002403), 0.24298794567584991),
        ((4.785215377807617, 0.0035884405951946974), (-0.0017961699049919844, 0.0011838842183351517),
         (-0.03521687909960747, 0.009459378197789192), 0.47957301139831543),
        ((4.727545261383057, 0.006344881374388933), (-0.0061436910182237625, 0.0008336332393810153),
         (0.012339762412011623, 0.0043235644698143005), 0.9500000476837158),
        ((4.806920528411865, 0.005436264909803867), (0.005690717604011297, 8.830774459056556e-5),
         (-0.00019663637795019895, 0.00024312522145919502), 0.949999988079071),
        ((4.726495742797852, 0.009544309228658676), (0.004016753751784563, 6.134989234851673e-5),
         (0.0006506261415779591, 0.00020928174490109086), 0.949999988079071),
        ((4.89390230178833, 0.0047211721539497375), (0.010379847139120102, 2.7608957680058666e-5),
         (0.00029396452009677887, 8.474134665448219e-5), 0.949999988079071),
        ((4.889120578765869, 0.002151205437257886), (0.0037524907384067774, 3.744014975382015e-5),
         (-0.0010508624836802483, 7.232622738229111e-5), 0.949999988079071),
        ((4.946272373199463, 0.008521423675119877), (0.001904668752104044, 5.143996168044396e-5),
         (-0.0012227826518937945, 7.035945600364357e-5), 0.949999988079071),
    ]
    for s, (w0, w1, w2, msd) in enumerate(want):
        for w, (m, v) in enumerate((w0, w1, w2)):
            assert st.mean[s, w] == m and st.var[s, w] == v
        assert st.msd[s] == msd
    assert st.gv_mean.tolist() == [0.03621548041701317]
    assert st.gv_var.tolist() == [0.00010934889724012464]
    assert st.gv_switch.tolist() == [0] * 5 + [1] * 30 + [0] * 5


DUR_10 = [8, 17, 14, 25, 15, 3, 4, 2, 2, 2, 2, 3, 3, 3, 3, 4, 3, 2, 2, 2, 3, 3, 6, 3, 2, 3,
          3, 3, 3, 2, 2, 1, 3, 2, 14, 22, 14, 26, 38, 5]
DUR_12 = [6, 12, 11, 19, 14, 3, 4, 2, 2, 2, 2, 3, 3, 3, 3, 4, 3, 2, 2, 2, 3, 3, 6, 3, 2, 3,
          3, 3, 3, 2, 2, 1, 3, 2, 14, 18, 11, 16, 27, 4]
DUR_AL = [36, 86, 48, 102, 27, 7, 11, 6, 6, 6, 2, 4, 3, 4, 3, 3, 3, 2, 2, 2, 3, 6, 14, 6, 3,
          4, 5, 6, 4, 3, 3, 1, 4, 4, 26, 28, 19, 42, 55, 8]
ALIGN = [(0.0, 298.5), (298.5, 334.5), (334.5, 350.5), (350.5, 362.5), (362.5, 394.5),
         (394.5, 416.5), (416.5, 454.5), (454.5, 606.5)]


def test_durations(oracle_voice):
    v = oracle_voice
    assert v.durations(SAMPLE_SENTENCE_1, 1.0).tolist() == DUR_10
    assert sum(DUR_10) == 277
    assert v.durations(SAMPLE_SENTENCE_1, 1.2).tolist() == DUR_12
    assert v.durations(SAMPLE_SENTENCE_1, 1.0, times=ALIGN).tolist() == DUR_AL


def test_label_alignment(oracle_voice):
    labels, times = oracle_voice.parse_label_lines(ALIGNED_1)
    assert labels == SAMPLE_SENTENCE_1
    np.testing.assert_allclose(times, np.array(ALIGN), rtol=4 * np.finfo(float).eps)


def test_mask_boundary_distances():
    bd = lambda m: [tuple(int(x) for x in p) for p in zip(*O.boundary_distances(m))]
    assert bd([1] * 10) == [(i, 9 - i) for i in range(10)]
    assert bd([1, 1, 1, 0, 0, 1, 1, 1, 1, 1]) == [
        (0, 2), (1, 1), (2, 0), (0, 0), (0, 0), (0, 4), (1, 3), (2, 2), (3, 1), (4, 0)]
    assert bd([1, 1, 1, 0, 1, 0, 0, 0, 0, 0]) == [(0, 2), (1, 1), (2, 0)] + [(0, 0)] * 7
    assert bd([]) == []


def test_bonsai(oracle_voice):
    s = oracle_voice.synthesize(SAMPLE_SENTENCE_1)
    assert len(s) == 66480
    assert abs(s[2000] - 19.35141137623778) <= EPS
    assert abs(s[30000] - -980.6757547598129) <= EPS


def test_is_this_bonsai(oracle_voice):
    s = oracle_voice.synthesize(SAMPLE_SENTENCE_2)
    assert len(s) == 100800
    assert abs(s[2000] - 17.15977345625943) <= EPS
    assert abs(s[30000] - 2566.2058730889985) <= EPS
    assert abs(s[70000] - -1898.2890228814217) <= EPS
    assert abs(s[100799] - -13.514971382534956) <= EPS


def test_is_this_bonsai_fast(oracle_voice):
    s = oracle_voice.synthesize(SAMPLE_SENTENCE_2, speed=1.4)
    assert len(s) == 72000
    assert abs(s[2000] - 15.0481014871396) <= EPS
    assert abs(s[30000] - -56.77163803227678) <= EPS
    assert abs(s[70000] - -9.15409432584658) <= EPS
    assert abs(s[71199] - 7.840225089163972) <= EPS


def test_bonsai_multi():
    """src/lib.rs:77-91 -- the reference's ONLY pin of VoiceSet::weighted on two different voices (BASELINE config
    5, SURVEY f-1).  Skipped until the tohoku-f01 files are supplied (tests/conftest.py: JB_TOHOKU_DIR)."""
    from tests.conftest import BONSAI_MULTI_GOLDEN, BONSAI_MULTI_WEIGHTS, tohoku_voices

    vs = O.VoiceSet(tohoku_voices(), BONSAI_MULTI_WEIGHTS)
    s = vs.synthesize(SAMPLE_SENTENCE_1)
    assert len(s) == BONSAI_MULTI_GOLDEN["len"]
    assert abs(s[2000] - BONSAI_MULTI_GOLDEN[2000]) <= EPS
    assert abs(s[30000] - BONSAI_MULTI_GOLDEN[30000]) <= EPS


def test_multiple_models():
    """src/model/mod.rs:395-428 -- the reference's second two-voice pin: blended duration pdfs and the first LF0 state
    of `Models` for tohoku neutral + happy, weights 0.7 / 0.3 on durations and on stream 1 (InterporationWeight::new's
    equal split elsewhere), BEFORE any arithmetic of the hot path.  assert_eq! in the reference: exact equality here.
    Skipped until the tohoku-f01 files are supplied (tests/conftest.py: JB_TOHOKU_DIR)."""
    from tests.conftest import (MULTIPLE_MODELS_DURATION, MULTIPLE_MODELS_LF0_STATE0, MULTIPLE_MODELS_WEIGHTS,
                                tohoku_voices)

    vs = O.VoiceSet(tohoku_voices(), MULTIPLE_MODELS_WEIGHTS)
    lab = [SAMPLE_SENTENCE_1[2]]
    assert vs.duration_params(lab).tolist() == [list(x) for x in MULTIPLE_MODELS_DURATION]
    st = vs.stream_states(1, lab)
    (w0, w1, w2), msd = MULTIPLE_MODELS_LF0_STATE0
    for w, (m, v) in enumerate((w0, w1, w2)):
        assert st.mean[0, w] == m and st.var[0, w] == v
    assert st.msd[0] == msd


def test_empty(oracle_voice):
    assert len(oracle_voice.synthesize([])) == 0
    assert len(oracle_voice.synthesize([], speed=1.2)) == 0


def test_stagewise_equals_end_to_end(oracle_voice):
    """The flat state-level entry (what the GPU boundary takes) reproduces the
    label-level synthesis bit for bit."""
    v = oracle_voice
    ref = v.synthesize(SAMPLE_SENTENCE_1, want_tracks=True)
    dur = v.durations(SAMPLE_SENTENCE_1)
    assert dur.tolist() == ref["dur"].tolist()
    tracks = [O.mlpg(v.stream_states(i, SAMPLE_SENTENCE_1), dur) for i in range(3)]
    assert np.array_equal(tracks[0], ref["mcp"])
    assert np.array_equal(tracks[1][:, 0], ref["lf0"])
    assert np.array_equal(tracks[2], ref["lpf"])
    pcm = O.vocoder(v.fs, v.fperiod, v.alpha, 1.0, tracks[1][:, 0], tracks[0], tracks[2])
    assert np.array_equal(pcm, ref["pcm"])


def test_postfilter_pieces():
    """X1 (beta > 0) has no golden in the reference's tests: PARITY UNPINNED.  These identities
    pin what can be pinned without one.
      * c2ir (cepstrum.rs:175-186): exp of a two-term series has the closed form e^c0 c1^n / n!
      * freqt (cepstrum.rs:153-173) with alpha = 0 is a shift register fed in ascending order, so
        the reference's input order REVERSES the coefficients (hts_engine feeds c1[m1]..c1[0]);
        with alpha != 0 it equals the textbook recursion applied to the reversed input
      * postfilter_mcp (cepstrum.rs:23-37): no-op for beta = 0 and for <= 2 coefficients; b[k>=2]
        scale by 1 + beta, b[1] -= beta*alpha*b[2]
    """
    import math

    ir = O.c2ir([0.3, 0.7], 12)
    want = [math.exp(0.3) * 0.7 ** n / math.factorial(n) for n in range(12)]
    np.testing.assert_allclose(ir, want, rtol=1e-14)
    assert np.array_equal(O.freqt([1.0, 2.0, 3.0], 5, 0.0), [3.0, 2.0, 1.0, 0.0, 0.0, 0.0])

    def textbook_freqt(c1, m2, a):
        g, d, aa = np.zeros(m2 + 1), np.zeros(m2 + 1), 1 - a * a
        for i in range(len(c1) - 1, -1, -1):
            d[0] = g[0]
            g[0] = c1[i] + a * d[0]
            d[1] = g[1]
            g[1] = aa * d[0] + a * g[1]
            for j in range(2, m2 + 1):
                d[j] = g[j]
                g[j] = d[j - 1] + a * (g[j] - g[j - 1])
        return g

    c = np.random.default_rng(7).normal(size=9)
    assert np.array_equal(O.freqt(c, 40, -0.55), textbook_freqt(c[::-1], 40, -0.55))

    a, beta = 0.55, 0.3
    mc = np.random.default_rng(8).normal(size=35) * 0.3
    assert np.array_equal(O.postfilter_mcp(mc, a, 0.0), mc)
    assert np.array_equal(O.postfilter_mcp(mc[:2], a, beta), mc[:2])

    def mc2b(cc):
        b = cc.copy()
        for i in range(len(cc) - 2, -1, -1):
            b[i] = cc[i] - a * b[i + 1]
        return b

    b0, b1 = mc2b(mc), mc2b(O.postfilter_mcp(mc, a, beta))
    np.testing.assert_allclose(b1[2:], b0[2:] * (1 + beta), rtol=1e-13)
    np.testing.assert_allclose(b1[1], b0[1] - beta * a * b0[2], rtol=1e-13)
    # the gain term: b[0] moves by ln(e1/e2)/2 with e from b2en (coefficients.rs:75-78)
    bm = b0.copy()
    bm[1] -= beta * a * bm[2]
    bm[2:] *= 1 + beta
    np.testing.assert_allclose(b1[0] - b0[0], math.log(O.b2en(b0, a) / O.b2en(bm, a)) / 2, rtol=1e-12)


def test_postfilter_changes_first_frame_only_gradually(oracle_voice):
    """vocoder/mod.rs:80-89: the first frame starts from the un-filtered coefficients, so with
    beta > 0 the very first samples are still those of the plain path's start."""
    v = oracle_voice
    r = v.synthesize(SAMPLE_SENTENCE_1, want_tracks=True)
    p1 = v.synthesize(SAMPLE_SENTENCE_1, beta=0.5)
    assert len(p1) == len(r["pcm"]) == 66480
    assert np.isfinite(p1).all()
    assert not np.allclose(p1, r["pcm"])


def test_native_build_same_bits():
    """bench.py's cpu_baseline times the oracle built -O3 -march=native (oracle/Makefile
    `native`).  Contraction stays off and there is no fast-math: it must give the checker build's bits, so
    the timed code is the pinned code."""
    from tests.golden.labels import SAMPLE_SENTENCE_1

    ref = O.Voice(VOICE).synthesize(SAMPLE_SENTENCE_1)
    try:
        O.use_library(O.build_native())
        got = O.Voice(VOICE).synthesize(SAMPLE_SENTENCE_1)
    finally:
        O.use_library(None)
    assert np.array_equal(got, ref)
    assert abs(got[30000] - -980.6757547598129) < 1e-10  # src/lib.rs:46
