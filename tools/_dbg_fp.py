import os, sys, dataclasses
sys.path.insert(0, os.getcwd())
import numpy as np
import jbonsai_amd as J
from jbonsai_amd import synth
from tests.conftest import VOICE
from tests.test_gpu_configs import oracle_pcm
from tests.helpers import rel_rms
import oracle.oracle as O
eng = J.Engine.load([VOICE]); tab = synth.VoiceTables(eng); vi = eng.voice_info()
u = synth.synth_utterance(tab, 500, 21)
vi2 = dataclasses.replace(vi, fperiod=90)
ref, tr = oracle_pcm(vi2, u)
r = O.vocoder(vi2.sampling_frequency, vi2.fperiod, vi2.alpha, 1.0, tr[1][:, 0], tr[0], tr[2], dumps=True)
pcm_o, exc_o, pulse_o = r
for name, kw in (("serial", dict(serial=True)), ("default", dict()), ("pair64", dict(chunk_frames=64, kernel="pair"))):
    with J.Batch(vi2, [u, u], keep_tracks=True, **kw) as b:
        b.run(); b.sync()
        g = [b.pcm(0), b.pcm(1)]; ex = [b.excitation(0), b.excitation(1)]
    for i in range(2):
        d = np.abs(ex[i] - exc_o); dp = np.abs(g[i] - pcm_o)
        print(name, "utt", i, "pcm rel rms", rel_rms(g[i], ref), "exc max diff", d.max(), "first exc diff", int(np.argmax(d > 1e-9)) if d.max() > 1e-9 else -1,
              "first pcm diff", int(np.argmax(dp > 1e-6 * np.abs(pcm_o).max())) if dp.max() > 1e-6*np.abs(pcm_o).max() else -1)
