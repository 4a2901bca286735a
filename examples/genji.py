"""The reference's `examples/genji` on the GPU path: the 1,456-label opening of Genji monogatari, TWO voices
interpolated 0.5 / 0.5 (duration, MCP, LF0; LPF from the first voice alone), labels -> PCM -> 16-bit WAV.

    python examples/genji.py [voice_a.htsvoice voice_b.htsvoice] [out.wav]

Mirrors examples/genji/main.rs of jbonsai: Engine::load([sad, happy]), the interpolation weights
(set_duration / set_parameter), Engine::synthesize.  The reference's voices (tohoku-f01-sad / -happy) are an
un-fetched submodule of its tree: without arguments the nitech voice and the permuted second voice of
tests/golden/make_permuted_voice.py stand in.  Also streams the same text through the SpeechGenerator.
Needs an MI355X: the library has no CPU path.
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import jbonsai_amd as J  # noqa: E402
from tests.golden.labels import GENJI  # examples/genji/genji.lab as a label list  # noqa: E402
from tests.golden.make_permuted_voice import NITECH, permuted_voice_path  # noqa: E402

args = [a for a in sys.argv[1:]]
out = args.pop() if args and args[-1].endswith(".wav") else "genji.wav"
with tempfile.TemporaryDirectory() as td:
    voices = args[:2] if len(args) >= 2 else [str(NITECH), str(permuted_voice_path(td))]
    engine = J.Engine.load(voices)
iw = engine.condition
iw.set_interpolation_duration([0.5, 0.5])
iw.set_interpolation_parameter(0, [0.5, 0.5])
iw.set_interpolation_parameter(1, [0.5, 0.5])
iw.set_interpolation_parameter(2, [1.0, 0.0])

t0 = time.perf_counter()
speech = engine.synthesize(GENJI)
dt = time.perf_counter() - t0
fs = engine.condition.get_sampling_frequency()
t0 = time.perf_counter()
speech = engine.synthesize(GENJI)  # again: the first call also initialised the device and filled the memory pools
dt2 = time.perf_counter() - t0
print(f"The synthesized voice has {len(speech)} samples in total "
      f"({len(speech) / fs:.1f} s of audio in {dt * 1e3:.1f} ms, {dt2 * 1e3:.1f} ms the second time).")
J.write_wav(out, speech, fs)
print(f"wrote {out}")

# the streaming iterator (src/speech.rs:65-96): 64 frames per call
g = engine.generator(GENJI)
buf = np.zeros(64 * g.fperiod())
t0 = time.perf_counter()
n = 0
first = None
while True:
    r = g.generate_steps(buf, 64)
    if r == 0:
        break
    first = first if first is not None else time.perf_counter() - t0
    n += r
print(f"streamed {n} samples; first 64 frames after {first * 1e3:.1f} ms, all after {(time.perf_counter() - t0) * 1e3:.1f} ms")
