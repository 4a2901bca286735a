"""The pulse walk in closed form (round 4, jb_vocoder.hip pulse_frame_closed; CPU model
tests/tools/pulse_closed_form.py) against the oracle's per-sample loop (src/vocoder/excitation.rs:25-33,73-81,
102-104) on pitch tracks that take every branch of it: periods that cross powers of two inside a frame (two
cur-segments), periods of a few samples (many fires per frame), jumps too large for a segment (samples taken the
loop's way), exact-tie increments (they come up once in fperiod frames by themselves: (p - p')/fperiod with p - p'
a multiple of the ulp), voiced runs of every length.  A pulse that is one sample off shows in the excitation as an
error of the pulse's size (~sqrt(period) * tap), nine orders above the gate."""
import numpy as np
import pytest

import jbonsai_amd as J
from oracle import oracle as O
from tests.golden.labels import SAMPLE_SENTENCE_2
from tests.helpers import voice_info

pytestmark = pytest.mark.gpu
NODATA = -1e10


def _tracks(v, T, lf0):
    r = v.synthesize(SAMPLE_SENTENCE_2, want_tracks=True)
    rep = -(-T // len(r["mcp"]))
    mcp = np.tile(r["mcp"], (rep, 1))[:T]
    lpf = np.tile(r["lpf"], (rep, 1))[:T]
    return J.TrackUtterance(mcp, lf0.reshape(T, 1), lpf)


def _lf0_tracks(T, rng):
    out = {}
    walk = 4.7 + np.cumsum(rng.normal(0, 0.01, T))
    out["smooth walk over 512 and 256"] = np.clip(walk, 4.2, 5.6)
    hi = 8.0 + np.cumsum(rng.normal(0, 0.02, T))
    out["periods of a few samples"] = np.clip(hi, 6.5, 9.9)  # ln 20000 = 9.903 is the clamp
    steps = np.repeat(rng.uniform(3.2, 9.0, T // 5 + 1), 5)[:T]
    out["jumps every five frames"] = steps
    saw = 4.5 + 0.9 * ((np.arange(T) % 40) / 40.0)
    out["sawtooth across binades"] = saw
    gaps = np.clip(walk, 4.2, 5.6).copy()
    k = 0
    while k < T:  # voiced runs of every length 1..60 with unvoiced gaps of 1..5
        run = 1 + (k * 7) % 60
        gaps[k + run:k + run + 1 + (k % 5)] = NODATA
        k += run + 1 + (k % 5)
    out["runs of every length"] = gaps
    out["constant pitch"] = np.full(T, np.log(48000.0 / 400.0))
    out["exactly 512 then down"] = np.concatenate([np.full(T // 2, np.log(48000.0 / 512.0)),
                                                   np.linspace(np.log(48000.0 / 512.0), np.log(48000.0 / 200.0), T - T // 2)])
    return out


@pytest.mark.parametrize("T", [3000, 6500])
def test_closed_form_pulse_walk_equals_the_loop(oracle_voice, T):
    v = oracle_voice
    vi = voice_info(v)
    rng = np.random.default_rng(T)
    cases = _lf0_tracks(T, rng)
    utts = [_tracks(v, T, lf0) for lf0 in cases.values()]
    with J.Batch(vi, utts, keep_tracks=True) as b:
        b.run()
        b.sync()
        for i, (name, lf0) in enumerate(cases.items()):
            u = utts[i]
            pcm, exc, pul = O.vocoder(vi.sampling_frequency, vi.fperiod, vi.alpha, 1.0, lf0, u.spectrum, u.lpf, dumps=True)
            got = b.excitation(i)
            bad = np.flatnonzero(np.abs(got - exc) > 1e-9)
            assert bad.size == 0, (name, bad[:5], got[bad[:5]], exc[bad[:5]], int((pul != 0).sum()))
            assert (pul != 0).sum() > 0 or "unvoiced" in name
