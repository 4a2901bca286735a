"""The shapes bench.py QUOTES numbers on, under the oracle (round 4; VERDICT r3 "next" 3):

  config4 / config5 records: 1024 distinct synthetic utterances x 6,386 frames created from pdf row indices
      (bench.py resident_record, ids 4000+i / 5000+i; config 5 over two different voices 0.5/0.5) -- four
      utterances of each against the oracle;
  config3_job: ONE pass of the 4096-utterance mixed-length list (bench.py config3_shard at N = 1: LPT into
      sub-batches of <= 7 M frames, created from pdf row indices): every utterance's length, eight utterances
      per sub-batch read back and finite, one per sub-batch against the oracle.
Tolerance: relative RMS <= PCM_TOL (tests/helpers.py) per utterance (north_star allows 1e-4), lengths exact."""
import numpy as np
import pytest

import jbonsai_amd as J
from jbonsai_amd import shard, synth
from tests.conftest import VOICE
from tests.golden.make_permuted_voice import permuted_voice_path
from tests.helpers import rel_rms, PCM_TOL
from tests.test_gpu_configs import oracle_pcm

pytestmark = pytest.mark.gpu
FRAMES = 6386       # bench.py CONFIG45_FRAMES
PICKS = (0, 341, 682, 1023)
SUB_BATCH_FRAMES = 7_000_000  # bench.py SUB_BATCH_FRAMES


def test_bench_constants_match():
    import bench

    assert bench.CONFIG45_FRAMES == FRAMES and bench.SUB_BATCH_FRAMES == SUB_BATCH_FRAMES


def test_config4_at_the_bench_shape():
    eng = J.Engine.load([VOICE])
    tab, vi = synth.VoiceTables(eng), eng.voice_info()
    pset = tab.pdf_set()
    utts = [synth.synth_utterance(tab, FRAMES, 4000 + i, indexed=True) for i in range(1024)]
    with J.Batch(vi, utts, pdf_set=pset) as b:
        b.run()
        b.sync()
        assert all(b.num_samples(i) == FRAMES * 240 for i in range(1024))
        got = {i: b.pcm(i) for i in PICKS}
        info = b.info()
    print("config 4 at the bench shape:", info)
    for i in PICKS:
        ref, _ = oracle_pcm(vi, synth.synth_utterance(tab, FRAMES, 4000 + i))
        assert len(got[i]) == len(ref) and rel_rms(got[i], ref) <= PCM_TOL, i
    pset.close()


def test_config5_at_the_bench_shape(tmp_path):
    eng = J.Engine.load([VOICE, permuted_voice_path(tmp_path)])
    tabs = [synth.VoiceTables(eng, 0), synth.VoiceTables(eng, 1)]
    half = {"duration": [0.5, 0.5], "parameter": [[0.5, 0.5]] * 3, "gv": [[0.5, 0.5]] * 3}
    vi = eng.voice_info()
    pset = synth.voice_set_pdf_set(tabs)
    utts = [synth.synth_utterance_voices(tabs, half, FRAMES, 5000 + i, indexed=True) for i in range(1024)]
    with J.Batch(vi, utts, pdf_set=pset) as b:
        b.run()
        b.sync()
        assert all(b.num_samples(i) == FRAMES * 240 for i in range(1024))
        got = {i: b.pcm(i) for i in PICKS}
    for i in PICKS:
        ref, _ = oracle_pcm(vi, synth.synth_utterance_voices(tabs, half, FRAMES, 5000 + i))
        assert len(got[i]) == len(ref) and rel_rms(got[i], ref) <= PCM_TOL, i
    pset.close()


def test_one_pass_of_the_config3_job():
    eng = J.Engine.load([VOICE])
    tab, vi = synth.VoiceTables(eng), eng.voice_info()
    pset = tab.pdf_set()
    lens = synth.mixed_lengths(4096)  # bench.py config3_shard at world size 1
    mine = shard.shard_for_rank(lens, 0, 1)
    assert sorted(mine) == list(range(4096))
    k = max(1, -(-sum(lens) // SUB_BATCH_FRAMES))
    subs = shard.lpt_partition([lens[i] for i in mine], k)
    assert sorted(j for sb in subs for j in sb) == list(range(4096))
    checked = 0
    for sb in subs:
        ids = [mine[j] for j in sb]
        utts = [synth.synth_utterance(tab, lens[i], 2000 + i, indexed=True) for i in ids]
        with J.Batch(vi, utts, pdf_set=pset) as b:
            b.run()
            b.sync()
            assert [b.num_samples(q) for q in range(len(ids))] == [lens[i] * 240 for i in ids]
            order = np.argsort([lens[i] for i in ids])
            look = list(order[:2]) + list(order[len(order) // 2:len(order) // 2 + 3]) + list(order[-3:])
            pcm = {int(q): b.pcm(int(q)) for q in look}
        for q, a in pcm.items():
            assert np.isfinite(a).all() and np.abs(a).max() > 1.0, ids[q]
        q = int(order[1])  # one of the shortest: the oracle takes ~0.15 s per 1000 frames
        ref, _ = oracle_pcm(vi, synth.synth_utterance(tab, lens[ids[q]], 2000 + ids[q]))
        assert len(pcm[q]) == len(ref) and rel_rms(pcm[q], ref) <= PCM_TOL, ids[q]
        checked += 1
    assert checked == len(subs) >= 8
    pset.close()
