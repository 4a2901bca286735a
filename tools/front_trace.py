"""Phases of the host front half (JB_FRONT_TRACE=1) for the bench's labels_to_pcm request (64 utterances x 1,500 labels)
and for one sentence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J
from tests.conftest import VOICE
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
eng = J.Engine.load([VOICE])
lb = [list(SAMPLE_SENTENCE_2) * 75] * 64
eng.synthesize_batch(lb, i16=True)
os.environ["JB_FRONT_TRACE"] = "1"
os.environ["JB_E2E_TIMING"] = "1"
print("== 64 x 1500 labels", flush=True)
eng.synthesize_batch(lb, i16=True)
print("== one utterance of 1500 labels", flush=True)
eng.synthesize_batch(lb[:1], i16=True)
print("== one sentence", flush=True)
eng.synthesize(SAMPLE_SENTENCE_1)
