"""Bitwise A/B aid: SHA-256 of the MCP/LF0 tracks and the PCM of fixed batches, to compare two builds of the
library (swap jbonsai_amd/libjbonsai_amd.so between runs): a mid-size batch of label utterances (wave
kernel), and 8 x the 25,546-frame synthetic utterance + 3 distinct ones (lane-triple kernel, resident GV,
LDS-staged band solve, split excitation: the kernels of BASELINE config 2), f64 and the 16-bit sink."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import jbonsai_amd as J  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2  # noqa: E402
from tests.helpers import oracle_states, to_utt, voice_info  # noqa: E402

v = O.Voice(VOICE)
utts = []
for lab, reps in ((SAMPLE_SENTENCE_2, 30), (SAMPLE_SENTENCE_1, 7), (SAMPLE_SENTENCE_2, 1)):
    d, s = oracle_states(v, list(lab) * reps)
    utts.append(to_utt(d, s))
with J.Batch(voice_info(v), utts * 3, keep_tracks=True) as b:
    b.run()
    b.sync()
    for name, arrs in (("mcp", [b.track(i, 0) for i in range(3)]), ("lf0", [b.track(i, 1) for i in range(3)]),
                       ("pcm", [b.pcm(i) for i in range(9)])):
        h = hashlib.sha256()
        for a in arrs:
            h.update(np.ascontiguousarray(a).tobytes())
        print(name, h.hexdigest()[:16])

from jbonsai_amd import synth  # noqa: E402

eng = J.Engine.load([VOICE])
tab, vi = synth.VoiceTables(eng), eng.voice_info()
big = [synth.u128(tab, 0)] * 8 + [synth.synth_utterance(tab, 9000 + 700 * i, 50 + i) for i in range(3)]
with J.Batch(vi, big, keep_tracks=True) as b:
    b.run()
    b.sync()
    print("config-2 kernels:", b.info())
    for name, arrs in (("mcp", [b.track(i, 0) for i in (0, 8, 10)]), ("lf0", [b.track(i, 1) for i in (0, 8, 10)]),
                       ("lpf", [b.track(i, 2) for i in (0, 9)]), ("exc", [b.excitation(i) for i in (0, 9)]),
                       ("pcm", [b.pcm(i) for i in (0, 7, 8, 9, 10)])):
        h = hashlib.sha256()
        for a in arrs:
            h.update(np.ascontiguousarray(a).tobytes())
        print("big", name, h.hexdigest()[:16])
with J.Batch(vi, big, pcm_i16=True) as b:
    b.run()
    b.sync()
    h = hashlib.sha256()
    for i in (0, 8, 10):
        h.update(b.pcm_i16(i).tobytes())
    print("big pcm_i16", h.hexdigest()[:16])
