"""The reference's two inner seams as first-class ABI entries:

  jb_mlpg_batch            MlpgAdjust::new(..).create(&durations) -> Vec<Vec<f64>>   src/mlpg_adjust/mod.rs:31-51
  jb_vocode_tracks_batch   SpeechGenerator::new + generate_all                      src/speech.rs:25-50,87-96
  jb_batch_create_from_tracks (+ jb_batch_run ...)  the same as a resident batch

The vocoder entry is fed the ORACLE's parameter tracks and compared with the oracle's vocoder on the same
tracks; the MLPG entry is fed the oracle's state-level inputs and compared with the oracle's MLPG."""
import numpy as np
import pytest

import jbonsai_amd as J
from jbonsai_amd import synth
from oracle import oracle as O
from tests.conftest import VOICE
from tests.golden.labels import BENCH_LETTER, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import oracle_states, rel_rms, to_utt, voice_info, PCM_TOL

pytestmark = pytest.mark.gpu
NODATA = -1e10


@pytest.fixture(scope="module")
def vi(oracle_voice):
    assert J.lib().jb_device_count() > 0
    return voice_info(oracle_voice)


def oracle_tracks(v, labels, **kw):
    r = v.synthesize(labels, want_tracks=True, **kw)
    return J.TrackUtterance(r["mcp"], r["lf0"], r["lpf"]), r["pcm"]


def test_vocode_tracks_batch_equals_oracle_vocoder(oracle_voice, vi):
    items = [oracle_tracks(oracle_voice, l) for l in (SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2, BENCH_LETTER)]
    empty = J.TrackUtterance(np.zeros((0, 35)), np.zeros((0, 1)), np.zeros((0, 31)))
    utts = [items[0][0], empty, items[1][0], items[2][0], items[0][0]]
    got = J.vocode_tracks_batch(vi, utts)
    assert [len(g) for g in got] == [66480, 0, 100800, 742 * 240, 66480]
    for g, (_, ref) in zip([got[0], got[2], got[3]], items):
        assert rel_rms(g, ref) <= PCM_TOL
    assert np.array_equal(got[0], got[4])
    assert abs(got[0][30000] - -980.6757547598129) < 1e-8  # src/lib.rs:46 through the tracks seam


def test_vocode_tracks_voicing_patterns(oracle_voice, vi):
    """Voicing comes from the lf0 track alone (NODATA = unvoiced, vocoder/mod.rs:73-77): all unvoiced, all
    voiced, single voiced frames, voiced first / last frame -- against the oracle's vocoder on the same tracks."""
    tu, _ = oracle_tracks(oracle_voice, SAMPLE_SENTENCE_2)
    T = len(tu.lf0)
    base = np.where(tu.lf0[:, 0] == NODATA, 5.0, tu.lf0[:, 0])
    pats = {
        "all unvoiced": np.full(T, NODATA),
        "all voiced": base.copy(),
        "alternating": np.where(np.arange(T) % 2 == 0, base, NODATA),
        "edges voiced": np.where((np.arange(T) < 1) | (np.arange(T) >= T - 1), base, NODATA),
        "runs of 7/3": np.where(np.arange(T) % 10 < 7, base, NODATA),
    }
    utts = [J.TrackUtterance(tu.spectrum, lf0, tu.lpf) for lf0 in pats.values()]
    got = J.vocode_tracks_batch(vi, utts)
    for (name, lf0), g in zip(pats.items(), got):
        ref = O.vocoder(vi.sampling_frequency, vi.fperiod, vi.alpha, 1.0, lf0, tu.spectrum, tu.lpf)
        assert rel_rms(g, ref) <= PCM_TOL, name


def test_vocode_tracks_mirrors_the_reference_panics(oracle_voice, vi):
    tu, _ = oracle_tracks(oracle_voice, SAMPLE_SENTENCE_1)
    T = len(tu.lf0)
    cases = [
        (J.TrackUtterance(tu.spectrum[:-1], tu.lf0, tu.lpf), "The length of spectrum, lf0, and lpf must be the same."),
        (J.TrackUtterance(tu.spectrum, tu.lf0, tu.lpf[:-2]), "The length of spectrum, lf0, and lpf must be the same."),
        (J.TrackUtterance(tu.spectrum, np.zeros((T, 2)), tu.lpf), "The size of lf0 static vector must be 1."),
        (J.TrackUtterance(tu.spectrum, tu.lf0, np.zeros((T, 30))),
         "The number of low-pass filter coefficient must be odd numbers."),
    ]
    for u, msg in cases:
        with pytest.raises(J.JbError) as ei:
            J.vocode_tracks_batch(vi, [tu, u])
        assert ei.value.code == -1 and msg in str(ei.value), (msg, str(ei.value))
    with pytest.raises(J.JbError):  # widths that are not the vocoder's (Vocoder::new nmcp / nlpf)
        J.vocode_tracks_batch(vi, [J.TrackUtterance(tu.spectrum[:, :30], tu.lf0, tu.lpf)])


def test_tracks_batch_resident_runs_and_options(oracle_voice, vi):
    """jb_batch_create_from_tracks as a resident batch: repeated runs, the 16-bit sink, the serial
    recursion and the post-filter, on the same tracks."""
    tu, ref = oracle_tracks(oracle_voice, SAMPLE_SENTENCE_2)
    with J.Batch(vi, [tu, tu]) as b:
        for _ in range(2):
            b.run()
            b.sync()
        a = b.pcm(0)
        assert np.array_equal(a, b.pcm(1)) and rel_rms(a, ref) <= PCM_TOL
        c = b.coefficients(0)
        assert c.shape == (len(tu.lf0), 35)
    with J.Batch(vi, [tu], serial=True) as b:
        b.run()
        b.sync()
        assert rel_rms(b.pcm(0), ref) <= 1e-12
    with J.Batch(vi, [tu], pcm_i16=True) as b:
        b.run()
        b.sync()
        assert np.array_equal(b.pcm_i16(0), np.clip(a, -32768.0, 32767.0).astype(np.int16))
    vb = voice_info(oracle_voice, beta=0.3)
    refb = O.vocoder(vi.sampling_frequency, vi.fperiod, vi.alpha, 1.0, tu.lf0[:, 0], tu.spectrum, tu.lpf, beta=0.3)
    assert rel_rms(J.vocode_tracks_batch(vb, [tu])[0], refb) <= PCM_TOL


def test_tracks_batch_at_scale(vi):
    """64 x 25,546 frames of tracks: the throughput kernel on caller-supplied tracks."""
    eng = J.Engine.load([VOICE])
    tab = synth.VoiceTables(eng)
    u = synth.u128(tab, 0)
    tr = J.mlpg_batch(vi, [u])[0]
    tu = J.TrackUtterance(tr[0], tr[1], tr[2])
    with J.Batch(vi, [u]) as b:
        b.run()
        b.sync()
        whole = b.pcm(0)
    with J.Batch(vi, [tu] * 64) as b:
        b.run()
        b.sync()
        info = b.info()
        got = [b.pcm(i) for i in (0, 63)]
    assert info["n_items"] >= 8192
    assert np.array_equal(got[0], got[1])
    assert rel_rms(got[0], whole) <= PCM_TOL


def test_mlpg_batch_equals_oracle_mlpg(oracle_voice, vi):
    sets = [oracle_states(oracle_voice, l) for l in (SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2, BENCH_LETTER)]
    utts = [to_utt(d, s) for d, s in sets]
    got = J.mlpg_batch(vi, utts)
    exact = J.mlpg_batch(vi, utts, serial_gv=True)
    for (dur, sts), g, x in zip(sets, got, exact):
        for s in range(3):
            ref = O.mlpg(sts[s], dur)
            assert g[s].shape == ref.shape
            np.testing.assert_allclose(g[s], ref, rtol=1e-12, atol=1e-13)
            if s > 0:
                assert np.array_equal(g[s], ref), s  # LF0 and LPF keep the reference's order: same bits
            assert np.array_equal(x[s], ref), s      # serial-order GV sums: MCP bit-exact as well
    # empty utterance, and tracks -> vocoder closes the loop with the one-call path
    e = J.mlpg_batch(vi, [to_utt(np.zeros(0, np.uint32), [type(s)(s.L, s.W, s.is_msd, s.use_gv, s.win_width, s.win_coef,
                                                                  np.zeros((0, s.L * s.W)), np.zeros((0, s.L * s.W)),
                                                                  np.zeros(0)) for s in sets[0][1]])])
    assert [t.shape for t in e[0]] == [(0, 35), (0, 1), (0, 31)]
    pcm = J.vocode_tracks_batch(vi, [J.TrackUtterance(*got[0])])[0]
    one = J.paramgen_vocode_batch(vi, [utts[0]])[0]
    assert rel_rms(pcm, one) <= 1e-10


def test_mlpg_only_batch_has_no_pcm(oracle_voice, vi):
    dur, sts = oracle_states(oracle_voice, SAMPLE_SENTENCE_1)
    with J.Batch(vi, [to_utt(dur, sts)], mlpg_only=True) as b:
        b.run()
        b.sync()
        assert np.array_equal(b.track(0, 1), O.mlpg(sts[1], dur))
        with pytest.raises(J.JbError):
            b.pcm(0)
        assert b.device_pcm()[0] is None


def test_tracks_entry_with_stage_nonzero_and_lsp_postfilter():
    """jb_vocode_tracks_batch for a Stage::NonZero vocoder (spectrum = [gain, LSP...], MGLSA cascade;
    vocoder/mod.rs:90-107,142-176): the tracks of a state-level batch, handed back through the tracks entry,
    give that batch's audio (same kernels behind the frame prologue; the conversion from the LSP track is
    ill-conditioned, so the comparison is the library against itself)."""
    from tests.test_gpu_stage import stable_utterance, stage_voice

    eng = J.Engine.load([VOICE])
    tab, vi0 = synth.VoiceTables(eng), eng.voice_info()
    for stage, log_gain, beta in ((2, False, 0.0), (3, False, 0.2)):
        v2 = stage_voice(vi0, stage, log_gain, beta)
        # random LSP sets can give an unstable filter: stable_utterance picks seeds whose output stays bounded
        utts = [stable_utterance(tab, vi0, v2, T, 900 + 10 * i, stage, log_gain, beta) for i, T in enumerate((300, 1100))]
        with J.Batch(v2, utts, keep_tracks=True) as b:
            b.run()
            b.sync()
            want = [b.pcm(i) for i in range(2)]
            trk = [[b.track(i, s) for s in range(3)] for i in range(2)]
        got = J.vocode_tracks_batch(v2, [J.TrackUtterance(*t) for t in trk])
        for g, w in zip(got, want):
            assert len(g) == len(w) and np.all(np.isfinite(g))
            assert rel_rms(g, w) <= 1e-10, (stage, rel_rms(g, w))
