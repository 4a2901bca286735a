"""The reference's `examples/is-bonsai` on the GPU path: full-context labels -> PCM -> 16-bit WAV.

    python examples/is_bonsai.py [voice.htsvoice] [out.wav]

Mirrors examples/is-bonsai/main.rs of jbonsai: Engine::load, Engine::synthesize, then the 16-bit mono
WAV the example writes with hound (clamp to i16, truncate).  Needs an MI355X: the library has no CPU path.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from tests.golden.labels import SAMPLE_SENTENCE_2  # the label lines of the reference's example  # noqa: E402

voice = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
    os.path.dirname(__file__), "..", "tests", "golden", "voice", "nitech_jp_atr503_m001.htsvoice")
out = sys.argv[2] if len(sys.argv) > 2 else "is-bonsai.wav"

engine = J.Engine.load([voice])
speech = engine.synthesize(SAMPLE_SENTENCE_2)
print(f"The synthesized voice has {len(speech)} samples in total.")
J.write_wav(out, speech, engine.condition.get_sampling_frequency())
print(f"wrote {out}")

# the batched entry with the 16-bit sink fused into the vocoder: four speeds of the same sentence
outs = []
for speed in (0.8, 1.0, 1.2, 1.4):
    engine.condition.set_speed(speed)
    outs.append(engine.synthesize_batch([SAMPLE_SENTENCE_2], i16=True)[0])
print("samples at speeds 0.8 / 1.0 / 1.2 / 1.4:", [len(o) for o in outs])
