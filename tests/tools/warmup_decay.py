"""Warm-up study (CPU): per start frame, the relative state error of a zero-started MLSA filter after w frames
(tests/tools/warmup_decay.c), for the config-2 utterance and a few distinct synthetic ones; prints the
distribution of the warm-up a hand-off needs and how simple per-frame features predict it."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402

so = ROOT / "oracle" / "build" / "warmup_decay.so"
src = Path(__file__).with_suffix(".c")
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"], check=True)
L = C.CDLL(str(so))
L.warmup_decay.argtypes = [C.c_int, C.c_double, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                           C.c_void_p, C.c_void_p]
VOICE = str(ROOT / "tests/golden/voice/nitech_jp_atr503_m001.htsvoice")
v = O.Voice(VOICE)
eng = J.Engine.load([VOICE]); tab = synth.VoiceTables(eng); vi = eng.voice_info()


def tracks(T, uid):
    u = synth.synth_utterance(tab, T, uid)
    sts = []
    for i, s in enumerate(u.streams):
        si = vi.streams[i]
        sts.append(O.StreamStates(si.vector_length, len(si.windows), si.is_msd, si.use_gv, [len(w) for w in si.windows],
                                  [c for w in si.windows for c in w], s.mean, s.var,
                                  s.msd if s.msd is not None else np.full(len(u.durations), 1.7976931348623157e308),
                                  s.gv_mean, s.gv_var, s.gv_switch))
    return [O.mlpg(s, u.durations) for s in sts]


def decay(tr, wmax=48, stride=1):
    pcm, exc, _ = O.vocoder(v.fs, v.fperiod, v.alpha, 1.0, tr[1][:, 0], tr[0], tr[2], dumps=True)
    T = len(tr[1])
    mcp = np.ascontiguousarray(tr[0]); exc = np.ascontiguousarray(exc)
    ns = (T + stride - 1) // stride
    out = np.zeros((ns, wmax)); smax = np.zeros(T + 1)
    rc = L.warmup_decay(v.fperiod, v.alpha, mcp.shape[1], T, mcp.ctypes.data, exc.ctypes.data, wmax, stride,
                        smax.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out, smax, mcp


def need(out, tol=1e-9):
    """need[t_out] = smallest w with err(t_out - w, w) <= tol (stride 1), inf if none up to wmax"""
    ns, wmax = out.shape
    nd = np.full(ns, np.inf)
    for w in range(wmax, 0, -1):
        t_out = np.arange(w, ns)
        ok = out[t_out - w, w - 1] <= tol
        nd[t_out[ok]] = w
    return nd


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    for uid in ([0] if T > 20000 else [0, 50, 51, 52]):
        tr = tracks(T, uid)
        out, smax, mcp = decay(tr)
        nd = need(out)[64:]
        fin = np.isfinite(nd)
        print(f"utt {uid} T={T}: needed warm-up percentiles (frames) 10/50/90/99/max:",
              np.percentile(nd[fin], [10, 50, 90, 99]), nd[fin].max(), "no w<=48:", (~fin).sum())
        for W in (8, 10, 12, 14, 16, 18, 20, 24, 32):
            print(f"   fixed W={W:2d}: {(nd > W).mean() * 100:6.2f} % of hand-off positions fail")
        np.save(f"/tmp/need_{uid}_{T}.npy", nd); np.save(f"/tmp/mcp_{uid}_{T}.npy", mcp); np.save(f"/tmp/lf0_{uid}_{T}.npy", tr[1])
        np.save(f"/tmp/smax_{uid}_{T}.npy", smax); np.save(f"/tmp/out_{uid}_{T}.npy", out)
