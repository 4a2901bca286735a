"""X1: mel-cepstral post-filter (beta > 0) on the GPU against the oracle's restatement of
MelCepstrum::postfilter_mcp (src/vocoder/cepstrum.rs:23-37; b2en coefficients.rs:75-78;
freqt cepstrum.rs:153-173; c2ir :175-186).

PARITY UNPINNED for this row: no reference test sets beta > 0, so the oracle's post-filter is
checked only by analytic identities (tests/test_oracle_golden.py::test_postfilter_pieces), not by
a golden vector.  Tolerances: filter coefficients abs 1e-12 (sums re-associated: 576-term
convolutions with FMAs, exp/log from the device library); PCM relative RMS <= 1e-9.
"""
import numpy as np
import pytest

import jbonsai_amd as J
from oracle import oracle as O
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import oracle_run, oracle_states, rel_rms, to_utt, voice_info, PCM_TOL

pytestmark = pytest.mark.gpu

COEF_TOL = 1e-12


def mc2b(c, a):
    b = np.array(c, dtype=np.float64, copy=True)
    for i in range(len(b) - 2, -1, -1):
        b[i] = c[i] - a * b[i + 1]
    return b


def oracle_coefficients(mcp, a, beta):
    return np.stack([mc2b(O.postfilter_mcp(row, a, beta), a) for row in mcp])


@pytest.mark.parametrize("beta", [0.1, 0.3, 1.0])
def test_coefficients_and_pcm(oracle_voice, beta):
    v = oracle_voice
    dur, sts = oracle_states(v, SAMPLE_SENTENCE_1)
    tracks, pcm = oracle_run(v, dur, sts, beta=beta)
    want = oracle_coefficients(tracks[0], v.alpha, beta)
    with J.Batch(voice_info(v, beta=beta), [to_utt(dur, sts)], keep_tracks=True) as b:
        b.run()
        b.sync()
        got = b.coefficients(0)
        # the parameter track itself is not touched (the filter lives inside the vocoder)
        np.testing.assert_allclose(b.track(0, 0), tracks[0], rtol=1e-12, atol=0)
        gpcm = b.pcm(0)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= COEF_TOL
    # b[k >= 2] is scaled by exactly 1 + beta (cepstrum.rs:29-31) up to the b2mc/mc2b round trip
    assert rel_rms(gpcm, pcm) <= PCM_TOL
    # the filter does change the output: not a no-op path
    assert rel_rms(gpcm, oracle_run(v, dur, sts)[1]) > 1e-2


def test_beta_zero_is_bitwise_the_plain_path(oracle_voice):
    v = oracle_voice
    dur, sts = oracle_states(v, SAMPLE_SENTENCE_1)
    a = J.paramgen_vocode_batch(voice_info(v), [to_utt(dur, sts)])[0]
    b = J.paramgen_vocode_batch(voice_info(v, beta=0.0), [to_utt(dur, sts)])[0]
    assert np.array_equal(a, b)


def test_first_frame_starts_unfiltered(oracle_voice):
    """Vocoder::synthesize seeds c with the UN-filtered mc2b(spectrum) on the first frame
    (vocoder/mod.rs:80-89), so frame 0 interpolates from there to the filtered target; chunked
    and serial schedules must agree on it."""
    v = oracle_voice
    dur, sts = oracle_states(v, SAMPLE_SENTENCE_2)
    _, pcm = oracle_run(v, dur, sts, beta=0.4)
    vi = voice_info(v, beta=0.4)
    with J.Batch(vi, [to_utt(dur, sts)], serial=True) as b:
        b.run()
        b.sync()
        ser = b.pcm(0)
    with J.Batch(vi, [to_utt(dur, sts)], chunk_frames=96) as b:
        b.run()
        b.sync()
        chk = b.pcm(0)
    fp = v.fperiod
    assert rel_rms(ser[:fp], pcm[:fp]) <= PCM_TOL and rel_rms(chk[:fp], pcm[:fp]) <= PCM_TOL
    assert rel_rms(ser, pcm) <= PCM_TOL and rel_rms(chk, pcm) <= PCM_TOL


def test_ragged_batch_with_empty(oracle_voice):
    v = oracle_voice
    d1, s1 = oracle_states(v, SAMPLE_SENTENCE_1)
    d2, s2 = oracle_states(v, SAMPLE_SENTENCE_2, speed=1.4)
    empty = to_utt(np.zeros(0, np.uint32), [type(s)(s.L, s.W, s.is_msd, s.use_gv, s.win_width, s.win_coef,
                                                      np.zeros((0, s.W * s.L)), np.zeros((0, s.W * s.L)),
                                                      np.zeros(0)) for s in s1])
    got = J.paramgen_vocode_batch(voice_info(v, beta=0.25), [to_utt(d1, s1), empty, to_utt(d2, s2)] * 3)
    r1, r2 = oracle_run(v, d1, s1, beta=0.25)[1], oracle_run(v, d2, s2, beta=0.25)[1]
    for k in range(3):
        assert len(got[3 * k + 1]) == 0
        assert rel_rms(got[3 * k], r1) <= PCM_TOL and rel_rms(got[3 * k + 2], r2) <= PCM_TOL
    assert np.array_equal(got[0], got[3]) and np.array_equal(got[2], got[8])


def test_many_frames_deterministic(oracle_voice):
    """More frames than waves in the grid (grid-stride path), identical rows give identical
    coefficients, and two runs are bitwise equal."""
    v = oracle_voice
    dur, sts = oracle_states(v, SAMPLE_SENTENCE_2)
    utts = [to_utt(dur, sts)] * 24  # 24 x 420 frames > 4096 waves
    vi = voice_info(v, beta=0.3)
    with J.Batch(vi, utts) as b:
        b.run()
        b.sync()
        c0, c23 = b.coefficients(0), b.coefficients(23)
        p0 = b.pcm(5)
        b.run()
        b.sync()
        assert np.array_equal(b.coefficients(23), c23) and np.array_equal(b.pcm(5), p0)
    assert np.array_equal(c0, c23)
    want = oracle_coefficients(O.mlpg(sts[0], dur), v.alpha, 0.3)
    assert np.max(np.abs(c0 - want)) <= COEF_TOL
