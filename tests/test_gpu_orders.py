"""The throughput vocoder kernel at every mel-cepstral order the reference takes (round 6; VERDICT r5 "missing" 1).

The reference is generic in the order (`/root/reference/src/vocoder/mod.rs:45-70`, `mlsa.rs:38-45,127-163`); until
round 6 `k_vocoder_lt` existed for nitech's two orders only and every other voice fell back to the wave-per-chunk
kernel.  Here: voices of other orders (synth.with_order: nitech's MCP stream cut or widened) through the C ABI --
 * small batches with the lane kernel forced, at the borders of every code (lane triples 25 / 31 / 35, one stage per
   lane 41 / 51 / 61) against the oracle;
 * full batches, 1024 utterances x 2,000 frames (eight distinct x 128 copies): `jb_batch_kernel_info` must report the
   lane kernel at two waves per SIMD, copies must be bitwise equal, the distinct ones meet the oracle."""
import numpy as np
import pytest

import jbonsai_amd as J
from jbonsai_amd import synth
from tests.conftest import VOICE
from tests.helpers import rel_rms, PCM_TOL
from tests.test_gpu_configs import oracle_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    eng = J.Engine.load([VOICE])
    return eng, synth.VoiceTables(eng), eng.voice_info()


@pytest.mark.parametrize("nmcp", [7, 12, 24, 26, 31, 32, 34, 36, 41, 42, 51, 52, 61])
def test_lane_kernel_small_batch_vs_oracle(ctx, nmcp):
    eng, tab, vi = ctx
    vi2, u2 = synth.with_order(vi, synth.synth_utterance(tab, 700, 3), nmcp)
    ref, _ = oracle_pcm(vi2, u2)
    assert np.isfinite(ref).all()
    for waves_hint in (64, 16):  # chunk lengths: few items (one wave per SIMD) / many
        with J.Batch(vi2, [u2, u2, u2], chunk_frames=waves_hint, kernel="triple") as b:
            b.run()
            b.sync()
            name, _ = b.kernel_info()
            got = [b.pcm(i) for i in range(3)]
        assert name == "k_vocoder_lt"
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
        assert rel_rms(got[0], ref) <= PCM_TOL, (nmcp, waves_hint)


@pytest.mark.parametrize("nmcp", [20, 30, 40, 50, 60])
def test_lane_kernel_full_batch(ctx, nmcp):
    eng, tab, vi = ctx
    T, distinct, copies = 2000, 8, 128
    pairs = [synth.with_order(vi, synth.synth_utterance(tab, T, 7000 + i), nmcp, seed=100 + i) for i in range(distinct)]
    vi2 = pairs[0][0]
    utts = [pairs[i % distinct][1] for i in range(distinct * copies)]
    with J.Batch(vi2, utts) as b:
        b.run()
        b.sync()
        name, waves = b.kernel_info()
        info = b.info()
        first = [b.pcm(i) for i in range(distinct)]
        for i in (distinct, 5 * distinct + 3, distinct * copies - 1):
            assert np.array_equal(b.pcm(i), first[i % distinct]), i
    print(f"order {nmcp - 1}: {name} at {waves} waves per SIMD, {info}")
    assert name == "k_vocoder_lt" and waves == 2
    for i in range(distinct):
        ref, _ = oracle_pcm(vi2, pairs[i][1])
        assert len(first[i]) == len(ref) and rel_rms(first[i], ref) <= PCM_TOL, (nmcp, i)
