"""Phases of jb_synthesize_batch (JB_E2E_TIMING=1) for the bench's labels_to_pcm request: 64 utterances x 158 s."""
import os, sys
os.environ["JB_E2E_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J
from tests.conftest import VOICE
from tests.golden.labels import SAMPLE_SENTENCE_2
eng = J.Engine.load([VOICE])
lb = [list(SAMPLE_SENTENCE_2) * 75] * 64
for i16 in (False, False, True, True):
    print("== i16" if i16 else "== f64", flush=True)
    eng.synthesize_batch(lb, i16=i16)
