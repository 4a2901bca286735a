"""Precision study (CPU, oracle only): f32 MLSA filter state vs the f64 oracle."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from oracle import oracle as O
from tests.golden.labels import SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2
from tests.helpers import rel_rms
import jbonsai_amd as J
from jbonsai_amd import synth

L = O.lib()
L.jbo_vocoder_f32state.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_size_t] + [C.c_void_p] * 5
VOICE = 'tests/golden/voice/nitech_jp_atr503_m001.htsvoice'
v = O.Voice(VOICE)


def study(dur, sts, name):
    tr = [O.mlpg(s, dur) for s in sts]
    pcm, exc, pul = O.vocoder(v.fs, v.fperiod, v.alpha, 1.0, tr[1][:, 0], tr[0], tr[2], dumps=True)
    out = np.zeros_like(pcm)
    lf0 = np.ascontiguousarray(tr[1][:, 0]); mcp = np.ascontiguousarray(tr[0]); lpf = np.ascontiguousarray(tr[2])
    L.jbo_vocoder_f32state(v.fs, v.fperiod, v.alpha, 1.0, 35, 31, len(lf0), lf0.ctypes.data, mcp.ctypes.data,
                           lpf.ctypes.data, exc.ctypes.data, out.ctypes.data)
    e = out - pcm
    print(name, "N", len(pcm), "rel RMS", rel_rms(out, pcm), "max abs", np.abs(e).max(),
          "sig rms", np.sqrt(np.mean(pcm ** 2)), "max", np.abs(pcm).max())


for lab, nm in ((SAMPLE_SENTENCE_1, "S1"), (SAMPLE_SENTENCE_2, "S2")):
    study(v.durations(lab), [v.stream_states(i, lab) for i in range(3)], nm)
eng = J.Engine.load([VOICE]); tab = synth.VoiceTables(eng); vi = eng.voice_info()
for T, uid in ((6000, 0), (6000, 7)):
    u = synth.synth_utterance(tab, T, uid)
    sts = []
    for i, s in enumerate(u.streams):
        si = vi.streams[i]
        sts.append(O.StreamStates(si.vector_length, len(si.windows), si.is_msd, si.use_gv, [len(w) for w in si.windows],
                                  [c for w in si.windows for c in w], s.mean, s.var,
                                  s.msd if s.msd is not None else np.full(len(u.durations), 1.7976931348623157e308),
                                  s.gv_mean, s.gv_var, s.gv_switch))
    study(u.durations, sts, f"synth{T}/{uid}")
