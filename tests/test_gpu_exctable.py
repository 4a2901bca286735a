"""Shared pulse-free excitation (round 4; jb_device.h VocDev::exc_tab / exc_src): where a voiced frame behind a
voiced frame carries the batch's canonical LPF taps, its excitation before the pulses is read from ONE table for
all utterances; everything else goes through the per-frame pass.  Both forms must give the same bits, for the
nitech voice (one low-pass filter: nearly every frame canonical) and for LPF taps that change from state to
state (nothing canonical but the first row, or a mixture), and match the oracle
(src/vocoder/excitation.rs:43-100)."""
import numpy as np
import pytest

import jbonsai_amd as J
from oracle import oracle as O
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import oracle_run, oracle_states, rel_rms, to_utt, voice_info, PCM_TOL

pytestmark = pytest.mark.gpu


def _perturbed(sts, mode, seed=5):
    """LPF means that differ from state to state: mode 'all' = every state its own taps, 'some' = every third
    state, 'runs' = stretches of 7 states alternate between the voice's taps and another set."""
    rng = np.random.default_rng(seed)
    lpf = sts[2]
    mean = np.array(lpf.mean, dtype=np.float64, copy=True)
    S = mean.shape[0]
    for s in range(S):
        hit = mode == "all" or (mode == "some" and s % 3 == 1) or (mode == "runs" and (s // 7) % 2 == 1)
        if hit:
            scale = 1.0 + 0.2 * rng.standard_normal(mean.shape[1]) if mode != "runs" else 0.8
            mean[s] = mean[s] * scale
    out = list(sts)
    out[2] = O.StreamStates(lpf.L, lpf.W, lpf.is_msd, lpf.use_gv, lpf.win_width, lpf.win_coef, mean, lpf.var, lpf.msd,
                            lpf.gv_mean, lpf.gv_var, lpf.gv_switch)
    return out


@pytest.mark.parametrize("mode", ["voice", "all", "some", "runs"])
def test_table_and_per_frame_pass_same_bits_and_oracle(oracle_voice, mode):
    v = oracle_voice
    utts, refs, excs = [], [], []
    for lab in (SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2):
        dur, sts = oracle_states(v, lab)
        if mode != "voice":
            sts = _perturbed(sts, mode)
        utts.append(to_utt(dur, sts))
        _, (pcm, exc, _) = oracle_run(v, dur, sts, dumps=True)
        refs.append(pcm)
        excs.append(exc)
    vi = voice_info(v)
    with J.Batch(vi, utts) as b:  # table where the taps are canonical
        b.run()
        b.sync()
        tab = [b.pcm(i) for i in range(2)]
    with J.Batch(vi, utts, no_exc_table=True) as b:  # every frame per utterance
        b.run()
        b.sync()
        gen = [b.pcm(i) for i in range(2)]
    with J.Batch(vi, utts, keep_tracks=True) as b:  # debug tap: the excitation itself
        b.run()
        b.sync()
        tap = [b.pcm(i) for i in range(2)]
        gexc = [b.excitation(i) for i in range(2)]
    for i in range(2):
        assert np.array_equal(tab[i], gen[i])
        assert np.array_equal(tab[i], tap[i])
        assert rel_rms(tab[i], refs[i]) <= PCM_TOL
        np.testing.assert_allclose(gexc[i], excs[i], rtol=0, atol=1e-9)


def test_table_path_in_the_throughput_kernel_and_ragged_batch(oracle_voice):
    """Many utterances of different lengths (the table is as long as the longest), the lane-triple kernel, the
    16-bit sink: same bits with and without the table."""
    v = oracle_voice
    eng = J.Engine.load([str(__import__("tests.conftest", fromlist=["VOICE"]).VOICE)])
    from jbonsai_amd import synth
    tab = synth.VoiceTables(eng)
    vi = eng.voice_info()
    utts = [synth.synth_utterance(tab, T, 300 + i) for i, T in enumerate([700, 1, 2500, 64, 1300, 65, 2499, 900])]
    for kw in ({}, {"kernel": "triple"}, {"pcm_i16": True}):
        outs = []
        for no_tab in (False, True):
            with J.Batch(vi, utts, no_exc_table=no_tab, **kw) as b:
                b.run()
                b.sync()
                outs.append([b.pcm_i16(i) if kw.get("pcm_i16") else b.pcm(i) for i in range(len(utts))])
        for a, c in zip(*outs):
            assert np.array_equal(a, c)
