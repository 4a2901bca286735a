import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J
from tests.conftest import VOICE
from tests.golden.labels import SAMPLE_SENTENCE_1
eng = J.Engine.load([VOICE])
for _ in range(5):
    eng.synthesize(SAMPLE_SENTENCE_1)
os.environ["JB_CREATE_TRACE"] = "1"
os.environ["JB_E2E_TIMING"] = "1"
os.environ["JB_FRONT_TRACE"] = "1"
for _ in range(3):
    eng.synthesize(SAMPLE_SENTENCE_1)
    print("--", file=sys.stderr)
