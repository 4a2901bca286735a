"""Round-2 GPU tests: multi-device entries, engine options against the oracle (half tone, alignment),
BASELINE config 2 / 3 at their per-GPU sizes, the certification fixes of the chunk hand-off
(re-certification after a full redo, carried-slot mask, repeated runs), read entries that order
behind the batch's streams, and the indexed (pdf row index) batch."""
import os

import numpy as np
import pytest

import jbonsai_amd as J
from jbonsai_amd import synth
from oracle import oracle as O
from tests.conftest import VOICE
from tests.golden.labels import ALIGNED_1, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import rel_rms, PCM_TOL, VERIFY_TOL
from tests.test_gpu_configs import oracle_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert J.lib().jb_device_count() > 0
    eng = J.Engine.load([VOICE])
    return eng, synth.VoiceTables(eng), eng.voice_info()


# ---- X3 and alignment mode through the engine entry, against the oracle -------------------------
@pytest.mark.parametrize("half_tone", [2.5, -7.0, 30.0])
def test_additional_half_tone_vs_oracle(oracle_voice, half_tone):
    """StreamParameter::apply_additional_half_tone (stream_parameter.rs:29-37; engine.rs:342-345):
    static LF0 mean += k*ln2/12, clamped to [ln 20, ln 20000] (30 half tones hit the upper clamp on
    some states), applied by the device gather; jb_synthesize vs the oracle's whole path."""
    e = J.Engine.load([VOICE])
    e.condition.set_additional_half_tone(half_tone)
    for labels in (SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2):
        got = e.synthesize(labels)
        ref = oracle_voice.synthesize(labels, half_tone=half_tone)
        assert got.shape == ref.shape
        err = rel_rms(got, ref)
        print("half tone", half_tone, len(labels), "labels: rel RMS vs oracle", err)
        assert err <= PCM_TOL
    # and it is not a no-op
    assert rel_rms(e.synthesize(SAMPLE_SENTENCE_1), oracle_voice.synthesize(SAMPLE_SENTENCE_1)) > 1e-3


def test_alignment_mode_vs_oracle(oracle_voice):
    """Condition::phoneme_alignment_flag with timed labels (label.rs:44, duration.rs:41-65): the
    durations come from the label times; jb_synthesize vs the oracle, also combined with a speed
    change and a half-tone shift (speed is ignored in alignment mode, duration.rs:28-38 vs :41)."""
    e = J.Engine.load([VOICE])
    e.condition.set_phoneme_alignment_flag(True)
    got = e.synthesize(ALIGNED_1)
    ref = oracle_voice.synthesize(ALIGNED_1, alignment=True)
    assert got.shape == ref.shape and len(got) != 66480  # not the free-running duration
    assert rel_rms(got, ref) <= PCM_TOL
    e.condition.set_additional_half_tone(-3.0)
    e.condition.set_speed(1.3)
    got = e.synthesize(ALIGNED_1)
    ref = oracle_voice.synthesize(ALIGNED_1, alignment=True, half_tone=-3.0, speed=1.3)
    assert got.shape == ref.shape
    assert rel_rms(got, ref) <= PCM_TOL


# ---- BASELINE config 2 and 3 at their per-GPU sizes ---------------------------------------------
def test_config2_at_batch_256(ctx):
    """BASELINE config 2 exactly as bench.py runs it: 256 copies of the 25,546-frame utterance.
    Copies are bitwise equal (each ran in a different lane triple / wave / CU), copies 0 and 255
    match the oracle, every hand-off is certified."""
    eng, tab, vi = ctx
    u = synth.u128(tab, 0)
    with J.Batch(vi, [u] * 256) as b:
        b.run()
        b.sync()
        info = b.info()
        picks = {i: b.pcm(i) for i in (0, 1, 100, 255)}
    # two waves per SIMD of the lane-triple kernel: 64 x 32 x 21 = 43,008 chunk slots
    assert 40000 <= info["n_items"] <= 43008 and 152 <= info["chunk_frames"] <= 160, info
    for i in (1, 100, 255):
        assert np.array_equal(picks[0], picks[i]), i
    ref, _ = oracle_pcm(vi, u)
    err = rel_rms(picks[0], ref)
    print("config 2 x256: rel RMS vs oracle", err, info)
    assert err <= PCM_TOL


def test_config3_per_gpu_share(ctx):
    """BASELINE config 3's share of one GPU of eight: 512 distinct utterances of the seed-fixed mixed
    lengths (rank 0's LPT share of the 4096 list), created from pdf row indices like bench.py's
    config-3 job; 8 of them against the oracle (shortest, longest and six spread between)."""
    from jbonsai_amd import shard

    eng, tab, vi = ctx
    lens = synth.mixed_lengths(4096)
    mine = shard.shard_for_rank(lens, 0, 8)
    assert len(mine) == 512
    pset = tab.pdf_set()
    utts = [synth.synth_utterance(tab, lens[i], 2000 + i, indexed=True) for i in mine]
    with J.Batch(vi, utts, pdf_set=pset) as b:
        b.run()
        b.sync()
        info = b.info()
        order = sorted(range(512), key=lambda j: lens[mine[j]])
        sample = [order[0], order[-1]] + [order[k] for k in (64, 128, 200, 300, 400, 480)]
        got = {j: b.pcm(j) for j in sample}
        assert b.total_samples == sum(lens[i] for i in mine) * 240
    print("config 3 share:", info)
    for j in sample:
        i = mine[j]
        u = synth.synth_utterance(tab, lens[i], 2000 + i)
        ref, _ = oracle_pcm(vi, u)
        assert len(got[j]) == lens[i] * 240
        err = rel_rms(got[j], ref)
        assert err <= PCM_TOL, (j, lens[i], err)
    pset.close()


def test_indexed_batch_equals_state_level_batch(ctx):
    """jb_batch_create_indexed (gather + blend on the device, one voice, weight 1) against
    jb_batch_create on the host-expanded arrays of the same utterances: same bits."""
    eng, tab, vi = ctx
    lens = [3, 77, 800, 2500]
    pset = tab.pdf_set()
    a = [synth.synth_utterance(tab, T, 300 + i) for i, T in enumerate(lens)]
    x = [synth.synth_utterance(tab, T, 300 + i, indexed=True) for i, T in enumerate(lens)]
    with J.Batch(vi, a, keep_tracks=True) as ba, J.Batch(vi, x, pdf_set=pset, keep_tracks=True) as bx:
        for b in (ba, bx):
            b.run()
            b.sync()
        for i in range(len(lens)):
            for s in range(3):
                assert np.array_equal(ba.track(i, s), bx.track(i, s)), (i, s)
            assert np.array_equal(ba.pcm(i), bx.pcm(i)), i
    pset.close()


# ---- multi-device entries (one GPU here: the device listed twice) -------------------------------
def test_paramgen_vocode_batch_multi_on_device_list(ctx):
    """jb_paramgen_vocode_batch_multi over devices = {0, 0}: LPT split by frames, one host thread
    per share, results in the caller's order; against the single-device entry (hand-off tolerance:
    the shares are different batches) and the oracle."""
    eng, tab, vi = ctx
    lens = [900, 40, 2200, 1, 650, 1500, 300, 0, 1200]
    utts = [synth.synth_utterance(tab, T, 700 + i) if T else
            J.Utterance(np.zeros(0, np.uint32), [J.StreamStates(np.zeros((0, 105)), np.zeros((0, 105))),
                                                 J.StreamStates(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0)),
                                                 J.StreamStates(np.zeros((0, 31)), np.zeros((0, 31)))])
            for i, T in enumerate(lens)]
    one = J.paramgen_vocode_batch(vi, utts, device=0)
    two = J.paramgen_vocode_batch(vi, utts, devices=[0, 0])
    three = J.paramgen_vocode_batch(vi, utts, devices=[0, 0, 0])
    for i, T in enumerate(lens):
        assert len(one[i]) == len(two[i]) == len(three[i]) == T * 240
        if T:
            assert rel_rms(two[i], one[i]) <= 1e-10 and rel_rms(three[i], one[i]) <= 1e-10, i
    ref, _ = oracle_pcm(vi, utts[2])
    assert rel_rms(two[2], ref) <= PCM_TOL
    with pytest.raises(J.JbError):
        J.paramgen_vocode_batch(vi, utts, devices=[0, 99])
    with pytest.raises(J.JbError):
        J.paramgen_vocode_batch(vi, utts, devices=[])


def test_synthesize_batch_multi_on_device_list(ctx):
    """jb_synthesize_batch_multi / _i16_multi over devices = {0, 0}: the reference's goldens come out
    of whichever share an utterance lands in; an error in one share surfaces with its message."""
    eng, tab, vi = ctx
    batch = [SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2, [], SAMPLE_SENTENCE_2 * 3, SAMPLE_SENTENCE_1] * 3
    one = eng.synthesize_batch(batch, device=0)
    two = eng.synthesize_batch(batch, devices=[0, 0])
    assert [len(o) for o in two] == [len(o) for o in one]
    for a, b in zip(one, two):
        if len(a):
            assert rel_rms(b, a) <= 1e-10
    assert abs(two[0][30000] - -980.6757547598129) < 1e-8     # src/lib.rs:46
    assert abs(two[1][70000] - -1898.2890228814217) < 1e-8    # src/lib.rs:131
    q = eng.synthesize_batch(batch, devices=[0, 0], i16=True)
    for a, b in zip(two, q):
        assert len(a) == len(b)
        if len(a):
            assert np.abs(b.astype(np.float64) - np.clip(a, -32768, 32767)).max() <= 1.0
    bad = list(SAMPLE_SENTENCE_1)
    bad[2] = "not a full-context label"
    with pytest.raises(J.JbError):
        eng.synthesize_batch(batch + [bad], devices=[0, 0])


# ---- chunk hand-off certification ----------------------------------------------------------------
@pytest.fixture(scope="module")
def long_utt(ctx):
    eng, tab, vi = ctx
    return vi, synth.synth_utterance(tab, 6000, 52)  # has slowly decaying stretches (tools/warmup_sweep.py)


def _serial(vi, u):
    with J.Batch(vi, [u], serial=True) as b:
        b.run()
        b.sync()
        return b.pcm(0)


@pytest.mark.parametrize("kern", ["wave", "triple"])
def test_recertification_after_full_redo(long_utt, kern):
    """16-frame chunks have no checkpoint: a failing chunk is recomputed to its end.  Its successor
    had been checked against the end state of the first pass -- a trajectory that started wrong and
    had only 16 + 6 frames to converge.  After the redo the successor is checked again against the
    exact end state and redone if it fails, so the hand-off tolerance bounds the error of every
    chunk: with verify_tol 1e-13 the result equals the serial recursion to 1e-12 (the throughput
    kernels' own summation order differs from the wave kernel's by 3e-14)."""
    vi, u = long_utt
    ser = _serial(vi, u)
    with J.Batch(vi, [u], chunk_frames=16, warmup_frames=6, verify_tol=1e-13, kernel=kern) as b:
        b.run()
        b.sync()
        info = b.info()
        out = b.pcm(0)
    err = rel_rms(out, ser)
    print(kern, "16-frame chunks, 6-frame warm-up, tol 1e-13:", info, "rel RMS vs serial", err)
    assert info["n_redo"] >= 50
    assert err <= 1e-12
    # the default tolerance bounds it as well, at its own level
    with J.Batch(vi, [u], chunk_frames=16, warmup_frames=6, verify_tol=VERIFY_TOL, kernel=kern) as b:
        b.run()
        assert rel_rms(b.pcm(0), ser) <= PCM_TOL


def test_repeated_runs_of_a_batch_with_failing_handoffs(ctx, long_utt):
    """A full redo (wave kernel) overwrites an end-state dump of the throughput kernel in place; the
    wave kernel also writes slots that are not carried state.  The hand-off check compares carried
    slots only, so running the same batch again finds the same failures and the same PCM."""
    eng, tab, vi = ctx
    _, u0 = long_utt
    utts = [u0] + [synth.synth_utterance(tab, 3000, 60 + i) for i in range(3)]
    for cf, wf in ((16, 6), (160, 6)):
        with J.Batch(vi, utts, chunk_frames=cf, warmup_frames=wf, verify_tol=VERIFY_TOL, kernel="triple") as b:
            seen = []
            for _ in range(3):
                b.run()
                b.sync()
                seen.append((b.info()["n_redo"], b.redo_stats(), [b.pcm(i) for i in range(len(utts))]))
        assert seen[0][0] > 0
        for k in (1, 2):
            assert seen[k][0] == seen[0][0] and seen[k][1] == seen[0][1], (cf, [s[:2] for s in seen])
            assert all(np.array_equal(a, c) for a, c in zip(seen[k][2], seen[0][2]))


def test_read_entries_order_behind_the_run(ctx):
    """jb_batch_read_* without an explicit jb_batch_sync: the read waits for the batch's (non-blocking)
    streams and for the certification + redo, so it returns the finished result."""
    eng, tab, vi = ctx
    utts = [synth.synth_utterance(tab, 4000, 90 + i) for i in range(8)]
    with J.Batch(vi, utts, chunk_frames=64, warmup_frames=4, verify_tol=VERIFY_TOL) as b:
        b.run()
        first = b.pcm(7)  # no sync() before
        assert b.info()["n_redo"] > 0
        b.run()
        b.sync()
        again = b.pcm(7)
    assert np.array_equal(first, again)
    ref, _ = oracle_pcm(vi, utts[7])
    assert rel_rms(first, ref) <= PCM_TOL


def test_staged_whole_slab_read(ctx):
    """jb_batch_read_pcm_all / _i16_all (the pinned staging ring of jb_synthesize_batch) against the
    per-utterance reads, ragged lengths including an empty and a one-frame utterance."""
    eng, tab, vi = ctx
    lens = [1200, 1, 5000, 333]
    utts = [synth.synth_utterance(tab, T, 40 + i) for i, T in enumerate(lens)]
    for i16 in (False, True):
        with J.Batch(vi, utts, pcm_i16=i16) as b:
            b.run()
            allp = b.pcm_all()
            for i in range(len(lens)):
                one = b.pcm_i16(i) if i16 else b.pcm(i)
                assert np.array_equal(allp[i], one), (i16, i)


def test_noise_table_growth_keeps_old_batches_valid(ctx):
    """The shared noise table grows when a longer utterance arrives; a batch created before keeps
    the table it was created with (shared ownership) and still produces the same PCM."""
    eng, tab, vi = ctx
    u = synth.synth_utterance(tab, 500, 11)
    with J.Batch(vi, [u]) as b:
        b.run()
        before = b.pcm(0)
        big = synth.synth_utterance(tab, 70000, 12)  # longer than anything the other tests create
        with J.Batch(vi, [big]) as b2:
            b2.run()
            n2 = len(b2.pcm(0))
        assert n2 == 70000 * 240
        b.run()
        assert np.array_equal(b.pcm(0), before)


def test_resident_gv_with_many_tiles_per_row(ctx):
    """k_mlpg_gv_gang with long rows: 70,000 frames = 18 workgroups per gang, and the limit case of
    32 (124,000 frames); rows longer than that take the multi-launch sweeps.  Tracks against the
    serial-order kernel (rtol 1e-12) -- every tile's record has to reach every other tile."""
    eng, tab, vi = ctx
    for T in (70000, 124000, 126000):
        u = synth.synth_utterance(tab, T, 77)
        res = []
        for kw in (dict(), dict(serial_gv=True)):
            with J.Batch(vi, [u], keep_tracks=True, **kw) as b:
                b.run()
                b.sync()
                res.append(b.track(0, 0))
        np.testing.assert_allclose(res[0], res[1], rtol=1e-12, atol=1e-13)


def test_gang_formation_timeout_falls_back_to_the_sweeps(ctx):
    """The resident GV kernel bounds its spins; with several such launches on one device a gang can time out
    in formation without any fault.  jb_batch_sync then redoes the step with the multi-launch GV sweeps (for
    that batch from then on) instead of failing.  JB_BATCH_TEST_GANG_TIMEOUT makes the first run behave that
    way; the error flag is sticky across run(); run(); sync()."""
    eng, tab, vi = ctx
    utts = [synth.synth_utterance(tab, T, 700 + i) for i, T in enumerate((3000, 1200, 2600))]
    with J.Batch(vi, utts, keep_tracks=True) as ref:
        ref.run()
        ref.sync()
        assert ref.gang_fallbacks() == 0
        want = [ref.pcm(i) for i in range(3)]
        wtrk = ref.track(0, 0)
    with J.Batch(vi, utts, keep_tracks=True, test_gang_timeout=True) as b:
        b.run()
        b.run()  # the first run's flag must survive the second launch
        b.sync()
        assert b.gang_fallbacks() == 1
        got = [b.pcm(i) for i in range(3)]
        np.testing.assert_allclose(b.track(0, 0), wtrk, rtol=1e-12, atol=1e-13)
        b.run()
        b.sync()
        assert b.gang_fallbacks() == 1  # the sweeps from now on
        again = [b.pcm(i) for i in range(3)]
    for g, a, w in zip(got, again, want):
        assert np.array_equal(g, a)
        assert rel_rms(g, w) <= 1e-10


def test_native_gather_world_of_one(ctx):
    """jb_comm_* / jb_gather_pcm (grouped ncclSend / ncclRecv over xGMI for N > 1).  One GPU here: a
    communicator of one rank gathers its own slab without a copy and without loading RCCL; that RCCL can be
    bound at run time is checked through jb_comm_unique_id.  N > 1 is the driver's to measure: RCCL refuses
    two ranks on one device, so it cannot be rehearsed on this box."""
    from jbonsai_amd import comm

    eng, tab, vi = ctx
    utts = [synth.synth_utterance(tab, T, 900 + i) for i, T in enumerate((700, 5, 1300))]
    assert len(comm.unique_id()) == 128  # librccl.so.1 is loadable and answers
    c = comm.Comm(None, 1, 0)
    for i16 in (False, True):
        with J.Batch(vi, utts, pcm_i16=i16) as b:
            b.run()  # not synced: the gather waits for the run and its certification
            g, ms = c.gather_pcm(b, root=0)
            assert g is not None and ms >= 0.0 and g.samples(0) == b.total_samples == 2005 * 240
            assert g.device_pointer(0) == b.device_pcm()[0]  # the root's own slab: no copy
            whole = g.read(0)
            for i in range(3):
                o = b.pcm_offset(i)
                part = b.pcm_i16(i) if i16 else b.pcm(i)
                assert np.array_equal(whole[o:o + len(part)], part)
            g.close()
    with pytest.raises(J.JbError):
        comm.Comm(None, 2, 0)  # more than one rank needs an id
    with pytest.raises(J.JbError):
        comm.Comm(None, 1, 3)
    c.close()
