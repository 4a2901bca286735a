"""Hand-off check failures of the time-chunked vocoder as a function of the warm-up length, over many
chunk geometries (= many different hand-off positions) of the config-2 utterance and of mixed
synthetic utterances.  Prints failures / hand-offs per warm-up."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402
from tests.conftest import VOICE  # noqa: E402

eng = J.Engine.load([VOICE])
tab = synth.VoiceTables(eng)
vi = eng.voice_info()
utts = [synth.u128(tab, 0)] + [synth.synth_utterance(tab, 6000, 50 + i) for i in range(3)]
for W in (16, 20, 24, 28, 32, 40):
    fails = handoffs = 0
    for ch in (96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192, 200, 208, 216, 224, 240):
        if ch < 2 * W:
            continue
        with J.Batch(vi, utts, chunk_frames=ch, warmup_frames=W, kernel="pair") as b:
            b.run()
            b.sync()
            info = b.info()
        fails += info["n_redo"]
        handoffs += info["n_items"] - len(utts)
    print(f"warm-up {W:2d} frames: {fails:5d} of {handoffs} hand-offs fail the 1e-9 check ({100.0 * fails / handoffs:.2f} %)", flush=True)
