import sys, time, numpy as np
sys.path.insert(0, '.')
import jbonsai_amd as J
from jbonsai_amd import synth
from tests.helpers import rel_rms
eng = J.Engine.load(["tests/golden/voice/nitech_jp_atr503_m001.htsvoice"])
tab = synth.VoiceTables(eng)
vi = eng.voice_info()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
t0=time.time(); u = synth.synth_utterance(tab, T, 0); print("gen", time.time()-t0, "S", len(u.durations), "T", int(u.durations.sum()))
vfrac = float(((u.streams[1].msd > 0.5) * u.durations).sum() / u.durations.sum()); print("voiced frac", vfrac)
t0=time.time(); b = J.Batch(vi, [u]*B); print("create", time.time()-t0)
for it in range(3):
    tot, voc = b.run_timed()
    ns = b.total_samples
    print(f"B={B} T={T}: total {tot:.2f} ms vocoder {voc:.2f} ms -> {ns/tot*1e3/1e6:.1f} Msamples/s ({ns/tot*1e3/48000:.0f}x RT); paramgen {tot-voc:.2f} ms")
p = b.pcm(0); print("pcm rms", np.sqrt(np.mean(p*p)), "max", np.abs(p).max(), "finite", np.isfinite(p).all())
if T <= 4000:
    from oracle import oracle as O
    ov = O.Voice("tests/golden/voice/nitech_jp_atr503_m001.htsvoice")
    from tests.helpers import voice_info
    sts = []
    for i, s in enumerate(u.streams):
        si = vi.streams[i]
        sts.append(O.StreamStates(si.vector_length, len(si.windows), si.is_msd, si.use_gv, [len(w) for w in si.windows], [c for w in si.windows for c in w], s.mean, s.var, s.msd if s.msd is not None else np.full(len(u.durations), 1.7976931348623157e308), s.gv_mean, s.gv_var, s.gv_switch))
    t0=time.time()
    tr = [O.mlpg(s, u.durations) for s in sts]
    ref = O.vocoder(48000, 240, 0.55, 1.0, tr[1][:,0], tr[0], tr[2]); dt=time.time()-t0
    print("oracle", dt, "s ->", len(ref)/dt/1e6, "Msamples/s; relrms", rel_rms(p, ref))
