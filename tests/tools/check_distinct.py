"""Parity spot check at scale: B distinct 128 s utterances through the default path, a few of them
against the oracle (each ~1.3 s of CPU).  Usage: python tests/tools/check_distinct.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.helpers import PCM_TOL  # noqa: E402
from tests.test_gpu_configs import oracle_pcm  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = J.Engine.load([VOICE])
tab, vi = synth.VoiceTables(eng), eng.voice_info()
utts = [synth.u128(tab, i) for i in range(B)]
with J.Batch(vi, utts) as b:
    b.run()
    b.sync()
    print("info", b.info(), "redo (settled, full)", b.redo_stats())
    for i in (0, B // 3, B - 1):
        ref, _ = oracle_pcm(vi, utts[i])
        got = b.pcm(i)
        e = float(np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
        print(f"utterance {i}: rel RMS vs oracle {e:.3e}")
        assert e <= PCM_TOL
print("ok")
