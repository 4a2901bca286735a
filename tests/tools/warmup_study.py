"""How fast does the MLSA filter forget its initial state?  (CPU, oracle only.)
For several start frames t0, run the f64 filter from ZERO state at t0 on the exact
excitation and compare with the full run: relative error per frame after t0."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from oracle import oracle as O
import jbonsai_amd as J
from jbonsai_amd import synth
from tests.golden.labels import SAMPLE_SENTENCE_2

L = O.lib()
L.jbo_vocoder_from_exc.argtypes = [C.c_int, C.c_double, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
VOICE = 'tests/golden/voice/nitech_jp_atr503_m001.htsvoice'
v = O.Voice(VOICE)
eng = J.Engine.load([VOICE]); tab = synth.VoiceTables(eng); vi = eng.voice_info()


def tracks_synth(T, uid):
    u = synth.synth_utterance(tab, T, uid)
    sts = []
    for i, s in enumerate(u.streams):
        si = vi.streams[i]
        sts.append(O.StreamStates(si.vector_length, len(si.windows), si.is_msd, si.use_gv, [len(w) for w in si.windows],
                                  [c for w in si.windows for c in w], s.mean, s.var,
                                  s.msd if s.msd is not None else np.full(len(u.durations), 1.7976931348623157e308),
                                  s.gv_mean, s.gv_var, s.gv_switch))
    return [O.mlpg(s, u.durations) for s in sts]


def study(tr, name, starts):
    pcm, exc, _ = O.vocoder(v.fs, v.fperiod, v.alpha, 1.0, tr[1][:, 0], tr[0], tr[2], dumps=True)
    T = len(tr[1]); fp = v.fperiod
    mcp = np.ascontiguousarray(tr[0])
    full = np.zeros_like(pcm)
    L.jbo_vocoder_from_exc(fp, v.alpha, 35, T, 0, mcp.ctypes.data, exc.ctypes.data, full.ctypes.data)
    print(name, "self-check vs oracle:", np.abs(full - pcm).max() / np.sqrt(np.mean(pcm ** 2)))
    sig = np.sqrt(np.mean(pcm ** 2))
    worst = {}
    for t0 in starts:
        out = np.zeros_like(pcm)
        L.jbo_vocoder_from_exc(fp, v.alpha, 35, T, t0, mcp.ctypes.data, exc.ctypes.data, out.ctypes.data)
        for W in (4, 8, 12, 16, 24, 32, 48, 64):
            a, b = (t0 + W) * fp, min(T, t0 + W + 40) * fp
            if a >= b:
                continue
            e = np.abs(out[a:b] - pcm[a:b]).max() / sig
            worst[W] = max(worst.get(W, 0.0), e)
    print(name, "max |err|/rms(signal) over 40 frames after a warm-up of W frames:")
    print("   ", "  ".join(f"W={W}: {e:.1e}" for W, e in sorted(worst.items())))


lab = SAMPLE_SENTENCE_2
tr = [O.mlpg(v.stream_states(i, lab), v.durations(lab)) for i in range(3)]
study(tr, "S2", [40, 100, 150, 200, 250, 300])
for uid in (0, 7, 11):
    study(tracks_synth(3000, uid), f"synth/{uid}", list(range(100, 2800, 173)))
