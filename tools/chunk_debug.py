import sys, numpy as np
sys.path.insert(0, '.')
import jbonsai_amd as J
from jbonsai_amd import synth
eng = J.Engine.load(["tests/golden/voice/nitech_jp_atr503_m001.htsvoice"])
tab = synth.VoiceTables(eng); vi = eng.voice_info()
T = int(sys.argv[1]) if len(sys.argv) > 1 else synth.T_128S
u = synth.synth_utterance(tab, T, 0)
def run(**kw):
    with J.Batch(vi, [u], **kw) as b:
        b.run(); b.sync()
        return b.pcm(0), b.info()
ser, _ = run(serial=True)
for W in (32, 48, 64):
    for tol in (1e-9, 1e30):
        out, info = run(chunk_frames=272, warmup_frames=W, verify_tol=tol)
        fp = 240
        nch = (T + 271) // 272
        errs = []
        sig = np.sqrt(np.mean(ser ** 2))
        for c in range(nch):
            a, b_ = c * 272 * fp, min(T, (c + 1) * 272) * fp
            errs.append(np.abs(out[a:b_] - ser[a:b_]).max() / sig)
        errs = np.array(errs)
        print(f"W={W} tol={tol:g}: n_redo={info['n_redo']} worst chunk err/rms={errs.max():.2e} at chunk {errs.argmax()} ; chunks>1e-12: {(errs>1e-12).sum()}")
