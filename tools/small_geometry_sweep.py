"""Chunk length x warm-up for SMALL requests (one sentence, a few sentences): time of run + sync (state-level batch,
inputs resident) and chunks redone, median of 15 runs.  The engine path uses the library's defaults (16 / 18)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jbonsai_amd as J
from jbonsai_amd import synth
from oracle import oracle as O  # only to turn labels into state-level inputs (host front half of the checker)
from tests.conftest import VOICE
from tests.golden.labels import BENCH_LETTER, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import oracle_states, to_utt, voice_info

v = O.Voice(VOICE)
vi = voice_info(v)
eng = J.Engine.load([VOICE]); tab = synth.VoiceTables(eng)
cases = {}
for name, lab in (("bonsai 277 fr", SAMPLE_SENTENCE_1), ("is_bonsai 420 fr", SAMPLE_SENTENCE_2), ("letter 742 fr", BENCH_LETTER)):
    d, s = oracle_states(v, list(lab))
    cases[name] = [to_utt(d, s)]
cases["8 synthetic x 400 fr"] = [synth.synth_utterance(tab, 400, 300 + i) for i in range(8)]
cases["synthetic 2000 fr"] = [synth.synth_utterance(tab, 2000, 77)]
for name, utts in cases.items():
    print("==", name)
    ref = None
    for ch in (16, 12, 8, 6, 4):
        row = []
        for wf in (18, 14, 12, 10, 8):
            with J.Batch(vi, utts, chunk_frames=ch, warmup_frames=wf) as b:
                for _ in range(3):
                    b.run(); b.sync()
                ts = []
                for _ in range(15):
                    t = time.perf_counter(); b.run(); b.sync(); ts.append(time.perf_counter() - t)
                pcm = b.pcm(0)
                if ref is None:
                    ref = pcm
                err = float(np.sqrt(np.mean((pcm - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
                row.append("%5.2f ms r%-2d %s" % (sorted(ts)[7] * 1e3, b.info()["n_redo"], "" if err < 1e-9 else "ERR %.1e" % err))
        print("chunk %2d | warm-up 18 14 12 10 8: " % ch + " | ".join(row))
