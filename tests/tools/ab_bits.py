"""Bitwise A/B aid: SHA-256 of the MCP/LF0 tracks and the PCM of a fixed mid-size batch, to compare two
builds of the library (swap jbonsai_amd/libjbonsai_amd.so between runs)."""
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
