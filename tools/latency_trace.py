"""Timeline (from rocprofv3 --kernel-trace) of one warm single-sentence synthesis: which kernels the
2.9 ms of GPU time are.  Usage: rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 tools/latency_trace.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from tests.conftest import VOICE  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_1  # noqa: E402

from tests.golden.labels import SAMPLE_SENTENCE_2  # noqa: E402

eng = J.Engine.load([VOICE])
reps = int(os.environ.get("JB_TRACE_REPS", "0"))  # 0: the 8-label sentence; n: n x the 20-label sentence (2.1 s each)
lab = list(SAMPLE_SENTENCE_2) * reps if reps else SAMPLE_SENTENCE_1
for _ in range(5):
    eng.synthesize(lab)
