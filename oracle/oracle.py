"""ctypes binding of the CPU oracle (oracle/jbo.h).

TEST INFRASTRUCTURE ONLY: may be imported from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from jbonsai_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "build" / "libjbo_oracle.so"

NODATA = -1e10
MAX_STREAM = 3


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (seconds)."""
    if force or not _LIB.exists() or any(
        p.stat().st_mtime > _LIB.stat().st_mtime for p in _HERE.glob("jbo*.[ch]")
    ):
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB


class Cond(C.Structure):
    _fields_ = [
        ("speed", C.c_double),
        ("volume", C.c_double),
        ("beta", C.c_double),
        ("additional_half_tone", C.c_double),
        ("msd_threshold", C.c_double * MAX_STREAM),
        ("gv_weight", C.c_double * MAX_STREAM),
        ("phoneme_alignment", C.c_int),
    ]


class Stream(C.Structure):
    _fields_ = [
        ("vector_length", C.c_uint32),
        ("num_windows", C.c_uint32),
        ("is_msd", C.c_uint32),
        ("use_gv", C.c_uint32),
        ("win_width", C.POINTER(C.c_uint32)),
        ("win_coef", C.POINTER(C.c_double)),
        ("mean", C.POINTER(C.c_double)),
        ("var", C.POINTER(C.c_double)),
        ("msd", C.POINTER(C.c_double)),
        ("gv_mean", C.POINTER(C.c_double)),
        ("gv_var", C.POINTER(C.c_double)),
        ("gv_switch", C.POINTER(C.c_uint8)),
        ("gv_weight", C.c_double),
        ("msd_threshold", C.c_double),
    ]


class Weights(C.Structure):
    _fields_ = [
        ("duration", C.POINTER(C.c_double)),
        ("parameter", C.POINTER(C.c_double) * MAX_STREAM),
        ("gv", C.POINTER(C.c_double) * MAX_STREAM),
    ]


_lib = None
_lib_path = None


def build_native() -> Path:
    """The -O3 -march=native build of the same sources, made on the machine that runs it (bench.py's
    cpu_baseline; oracle/Makefile `native`)."""
    subprocess.run(["make", "-C", str(_HERE), "native"], check=True, capture_output=True)
    return _HERE / "build" / "libjbo_oracle_native.so"


def use_library(path=None):
    """Bind the module to another build of the oracle (None: the default checker build)."""
    global _lib, _lib_path
    _lib, _lib_path = None, (None if path is None else Path(path))
    return lib()


def lib():
    global _lib
    if _lib is None:
        if _lib_path is None:
            build()
        L = C.CDLL(str(_lib_path or _LIB))
        L.jbo_voice_load.restype = C.c_void_p
        L.jbo_voice_load.argtypes = [C.c_char_p]
        L.jbo_voice_free.argtypes = [C.c_void_p]
        for n in ("sampling_frequency", "fperiod", "nstate", "nstream", "stage"):
            f = getattr(L, "jbo_voice_" + n)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p]
        L.jbo_voice_alpha.restype = C.c_double
        L.jbo_voice_alpha.argtypes = [C.c_void_p]
        for n in ("vector_length", "num_windows", "is_msd", "use_gv"):
            f = getattr(L, "jbo_voice_" + n)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int]
        L.jbo_voice_window.restype = C.c_int
        L.jbo_voice_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.jbo_voice_ntree.restype = C.c_int
        L.jbo_voice_ntree.argtypes = [C.c_void_p, C.c_int]
        L.jbo_voice_npdf.restype = C.c_int
        L.jbo_voice_npdf.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.jbo_voice_pdf_len.restype = C.c_int
        L.jbo_voice_pdf_len.argtypes = [C.c_void_p, C.c_int]
        L.jbo_voice_pdf.restype = C.POINTER(C.c_float)
        L.jbo_voice_pdf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.jbo_voice_get_index.restype = C.c_int
        L.jbo_voice_get_index.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.jbo_cond_default.argtypes = [C.POINTER(Cond)]
        L.jbo_duration_params.restype = C.c_int
        L.jbo_duration_params.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_void_p]
        L.jbo_durations.restype = C.c_int
        L.jbo_durations.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_double,
                                    C.c_void_p, C.c_void_p]
        L.jbo_stream_params.restype = C.c_int
        L.jbo_stream_params.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.jbo_gv_params.restype = C.c_int
        L.jbo_gv_params.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.jbo_parse_label_lines.restype = C.c_int
        L.jbo_parse_label_lines.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                            C.POINTER(C.c_char_p), C.c_void_p]
        L.jbo_mask.restype = C.c_size_t
        L.jbo_mask.argtypes = [C.POINTER(Stream), C.c_uint32, C.c_void_p, C.c_void_p]
        L.jbo_boundary_distances.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.jbo_mlpg.restype = C.c_int
        L.jbo_mlpg.argtypes = [C.POINTER(Stream), C.c_uint32, C.c_void_p, C.c_void_p]
        L.jbo_vocoder.restype = C.c_int
        L.jbo_vocoder.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                  C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
        L.jbo_vocoder_beta.restype = C.c_int
        L.jbo_vocoder_beta.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                       C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        L.jbo_freqt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double]
        L.jbo_c2ir.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.jbo_b2en.restype = C.c_double
        L.jbo_b2en.argtypes = [C.c_void_p, C.c_size_t, C.c_double]
        L.jbo_postfilter_mcp.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double]
        L.jbo_noise.argtypes = [C.c_void_p, C.c_size_t]
        vp, sz, db = C.c_void_p, C.c_size_t, C.c_double
        L.jbo_lsp2lpc.argtypes = [vp, sz, vp]
        L.jbo_gnorm.argtypes = [vp, sz, db]
        L.jbo_ignorm.argtypes = [vp, sz, db]
        L.jbo_gc2gc.argtypes = [vp, sz, db, vp, sz, db]
        L.jbo_mgc2mgc.argtypes = [vp, sz, db, db, vp, sz, db, db]
        L.jbo_lsp2mgc.argtypes = [vp, sz, db, C.c_int, sz, db, vp]
        L.jbo_postfilter_lsp.argtypes = [vp, sz, db, C.c_int, sz, db, db]
        L.jbo_check_lsp_stability.argtypes = [vp, sz]
        L.jbo_stage_coefficients.argtypes = [vp, sz, db, db, C.c_int, sz, C.c_int, vp]
        L.jbo_mglsa_df.argtypes = [vp, sz, sz, vp, db, vp]
        L.jbo_vocoder_stage.restype = C.c_int
        L.jbo_vocoder_stage.argtypes = [C.c_int, C.c_int, db, db, db, C.c_int, C.c_int, C.c_int, C.c_int, sz,
                                        vp, vp, vp, vp, vp]
        L.jbo_vocoder_stage_coef.restype = C.c_int
        L.jbo_vocoder_stage_coef.argtypes = [C.c_int, C.c_int, db, db, db, C.c_int, C.c_int, C.c_int, C.c_int, sz,
                                             vp, vp, vp, vp, vp, vp, vp]
        L.jbo_synthesize_ex.restype = C.c_int
        L.jbo_synthesize_ex.argtypes = [C.c_void_p, C.POINTER(Cond), C.POINTER(C.c_char_p), C.c_int,
                                        C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t),
                                        C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32),
                                        C.POINTER(C.POINTER(C.c_double)),
                                        C.POINTER(C.POINTER(C.c_double)),
                                        C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t)]
        vpp, dpp = C.POINTER(C.c_void_p), C.POINTER(C.c_double)
        L.jbo_duration_params_multi.restype = C.c_int
        L.jbo_duration_params_multi.argtypes = [vpp, C.c_int, dpp, C.POINTER(C.c_char_p), C.c_int, C.c_void_p]
        L.jbo_durations_multi.restype = C.c_int
        L.jbo_durations_multi.argtypes = [vpp, C.c_int, dpp, C.POINTER(C.c_char_p), C.c_int, C.c_double,
                                          C.c_void_p, C.c_void_p]
        L.jbo_stream_params_multi.restype = C.c_int
        L.jbo_stream_params_multi.argtypes = [vpp, C.c_int, dpp, C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.jbo_gv_params_multi.restype = C.c_int
        L.jbo_gv_params_multi.argtypes = [vpp, C.c_int, dpp, C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.jbo_synthesize_multi_ex.restype = C.c_int
        L.jbo_synthesize_multi_ex.argtypes = [vpp, C.c_int, C.POINTER(Weights)] + L.jbo_synthesize_ex.argtypes[1:]
        L.jbo_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _strs(labels):
    arr = (C.c_char_p * max(1, len(labels)))()
    for i, s in enumerate(labels):
        arr[i] = s.encode() if isinstance(s, str) else s
    return arr


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


class StreamStates:
    """Flat state-level arrays of one stream (image of ModelStream)."""

    def __init__(self, L, W, is_msd, use_gv, win_width, win_coef, mean, var, msd,
                 gv_mean=None, gv_var=None, gv_switch=None, gv_weight=1.0, msd_threshold=0.5):
        self.L, self.W, self.is_msd, self.use_gv = int(L), int(W), int(is_msd), int(use_gv)
        self.win_width = np.ascontiguousarray(win_width, dtype=np.uint32)
        self.win_coef = np.ascontiguousarray(win_coef, dtype=np.float64)
        self.mean = np.ascontiguousarray(mean, dtype=np.float64)
        self.var = np.ascontiguousarray(var, dtype=np.float64)
        self.msd = np.ascontiguousarray(msd, dtype=np.float64)
        self.gv_mean = None if gv_mean is None else np.ascontiguousarray(gv_mean, dtype=np.float64)
        self.gv_var = None if gv_var is None else np.ascontiguousarray(gv_var, dtype=np.float64)
        self.gv_switch = None if gv_switch is None else np.ascontiguousarray(gv_switch, dtype=np.uint8)
        self.gv_weight, self.msd_threshold = float(gv_weight), float(msd_threshold)

    def c_struct(self) -> Stream:
        s = Stream()
        s.vector_length, s.num_windows, s.is_msd, s.use_gv = self.L, self.W, self.is_msd, self.use_gv
        s.win_width = _ptr(self.win_width, C.c_uint32)
        s.win_coef = _ptr(self.win_coef, C.c_double)
        s.mean = _ptr(self.mean, C.c_double)
        s.var = _ptr(self.var, C.c_double)
        s.msd = _ptr(self.msd, C.c_double)
        s.gv_mean = _ptr(self.gv_mean, C.c_double)
        s.gv_var = _ptr(self.gv_var, C.c_double)
        s.gv_switch = _ptr(self.gv_switch, C.c_uint8)
        s.gv_weight, s.msd_threshold = self.gv_weight, self.msd_threshold
        return s


class Voice:
    def __init__(self, path):
        self.L = lib()
        self.h = self.L.jbo_voice_load(str(path).encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load voice {path}")
        g = self.L
        self.fs = g.jbo_voice_sampling_frequency(self.h)
        self.fperiod = g.jbo_voice_fperiod(self.h)
        self.nstate = g.jbo_voice_nstate(self.h)
        self.nstream = g.jbo_voice_nstream(self.h)
        self.alpha = g.jbo_voice_alpha(self.h)
        self.stage = g.jbo_voice_stage(self.h)
        self.vector_length = [g.jbo_voice_vector_length(self.h, i) for i in range(self.nstream)]
        self.num_windows = [g.jbo_voice_num_windows(self.h, i) for i in range(self.nstream)]
        self.is_msd = [g.jbo_voice_is_msd(self.h, i) for i in range(self.nstream)]
        self.use_gv = [g.jbo_voice_use_gv(self.h, i) for i in range(self.nstream)]
        self.windows = []
        for i in range(self.nstream):
            ws = []
            for w in range(self.num_windows[i]):
                buf = (C.c_double * 16)()
                n = g.jbo_voice_window(self.h, i, w, buf, 16)
                ws.append([buf[k] for k in range(n)])
            self.windows.append(ws)

    def __del__(self):
        try:
            if self.h:
                self.L.jbo_voice_free(self.h)
                self.h = None
        except Exception:
            pass

    def get_index(self, kind, state_index, label):
        ts, pi = C.c_int(), C.c_int()
        r = self.L.jbo_voice_get_index(self.h, kind, state_index, label.encode(), C.byref(ts), C.byref(pi))
        return (None if r or ts.value < 0 else ts.value, None if r else pi.value)

    def pdf_table(self, kind, tree):
        n = self.L.jbo_voice_npdf(self.h, kind, tree)
        pl = self.L.jbo_voice_pdf_len(self.h, kind)
        p = self.L.jbo_voice_pdf(self.h, kind, tree, 1)
        return np.ctypeslib.as_array(p, shape=(n, pl)).copy()

    def ntree(self, kind):
        return self.L.jbo_voice_ntree(self.h, kind)

    def duration_params(self, labels):
        S = len(labels) * self.nstate
        out = np.zeros((S, 2))
        if self.L.jbo_duration_params(self.h, _strs(labels), len(labels), out.ctypes.data):
            raise RuntimeError("duration_params")
        return out

    def durations(self, labels, speed=1.0, times=None):
        S = len(labels) * self.nstate
        out = np.zeros(S, dtype=np.uint32)
        t = None
        if times is not None:
            t = np.ascontiguousarray(times, dtype=np.float64)
        if self.L.jbo_durations(self.h, _strs(labels), len(labels), float(speed),
                                t.ctypes.data if t is not None else None, out.ctypes.data):
            raise RuntimeError("durations")
        return out

    def parse_label_lines(self, lines):
        n = len(lines)
        keep = _strs(lines)
        outp = (C.c_char_p * max(1, n))()
        times = np.zeros((max(1, n), 2))
        m = self.L.jbo_parse_label_lines(self.fs, self.fperiod, keep, n, outp, times.ctypes.data)
        if m < 0:
            raise RuntimeError("label parse")
        return [outp[i].decode() for i in range(m)], times[:m].copy()

    def stream_states(self, stream, labels, gv_weight=1.0, msd_threshold=0.5) -> StreamStates:
        S = len(labels) * self.nstate
        L, W = self.vector_length[stream], self.num_windows[stream]
        mean = np.zeros((S, W * L))
        var = np.zeros((S, W * L))
        msd = np.zeros(S)
        if self.L.jbo_stream_params(self.h, stream, _strs(labels), len(labels), mean.ctypes.data,
                                    var.ctypes.data, msd.ctypes.data):
            raise RuntimeError("stream_params")
        gm = gv = gs = None
        if self.use_gv[stream] and len(labels):
            gm, gv, gs = np.zeros(L), np.zeros(L), np.zeros(S, dtype=np.uint8)
            if self.L.jbo_gv_params(self.h, stream, _strs(labels), len(labels), gm.ctypes.data,
                                    gv.ctypes.data, gs.ctypes.data):
                raise RuntimeError("gv_params")
        ww = [len(w) for w in self.windows[stream]]
        wc = [c for w in self.windows[stream] for c in w]
        return StreamStates(L, W, self.is_msd[stream], self.use_gv[stream], ww, wc, mean, var, msd,
                            gm, gv, gs, gv_weight, msd_threshold)

    def synthesize(self, lines, speed=1.0, volume=1.0, half_tone=0.0, alignment=False,
                   gv_weight=None, msd_threshold=None, want_tracks=False, beta=0.0):
        c = Cond()
        self.L.jbo_cond_default(C.byref(c))
        c.speed, c.volume, c.additional_half_tone = speed, volume, half_tone
        c.beta = beta
        c.phoneme_alignment = int(alignment)
        if gv_weight is not None:
            for i, x in enumerate(gv_weight):
                c.gv_weight[i] = x
        if msd_threshold is not None:
            for i, x in enumerate(msd_threshold):
                c.msd_threshold[i] = x
        pcm = C.POINTER(C.c_double)()
        n = C.c_size_t()
        dur = C.POINTER(C.c_uint32)()
        S = C.c_uint32()
        mcp = C.POINTER(C.c_double)()
        lf0 = C.POINTER(C.c_double)()
        lpf = C.POINTER(C.c_double)()
        T = C.c_size_t()
        r = self.L.jbo_synthesize_ex(self.h, C.byref(c), _strs(lines), len(lines), C.byref(pcm),
                                     C.byref(n), C.byref(dur), C.byref(S), C.byref(mcp),
                                     C.byref(lf0), C.byref(lpf), C.byref(T))
        if r:
            raise RuntimeError(f"oracle synthesize failed: {r}")

        def take(p, shape):
            if not p or int(np.prod(shape)) == 0:
                return np.zeros(shape)
            a = np.ctypeslib.as_array(p, shape=shape).copy()
            return a

        out = take(pcm, (n.value,))
        res = out
        if want_tracks:
            Tn = T.value
            res = dict(pcm=out, T=Tn,
                       dur=(np.ctypeslib.as_array(dur, shape=(S.value,)).copy() if dur else np.zeros(0, np.uint32)),
                       mcp=take(mcp, (Tn, self.vector_length[0])),
                       lf0=take(lf0, (Tn,)),
                       lpf=take(lpf, (Tn, self.vector_length[2] if self.nstream > 2 else 0)))
        for p in (pcm, dur, mcp, lf0, lpf):
            if p:
                self.L.jbo_free(p)
        return res


class VoiceSet:
    """Several voices + InterporationWeight: VoiceSet::weighted (src/model/voice_set.rs:80-95) in front of
    the single-voice path.  weights: {"duration": [nv], "parameter": [[nv] per stream], "gv": [[nv] per
    stream]}; missing entries are the equal weights of InterporationWeight::new
    (src/model/interporation_weight.rs:48-60).  PARITY UNPINNED for two different voices (the reference's
    `bonsai_multi` golden needs voice files that are not in its tree)."""

    def __init__(self, paths, weights=None):
        self.voices = [Voice(p) for p in paths]
        self.v0 = self.voices[0]
        self.L = lib()
        nv = len(self.voices)
        eq = [1.0 / nv] * nv
        w = dict(weights or {})
        self.w_duration = np.ascontiguousarray(w.get("duration", eq), dtype=np.float64)
        self.w_parameter = [np.ascontiguousarray(x, dtype=np.float64)
                            for x in w.get("parameter", [eq] * self.v0.nstream)]
        self.w_gv = [np.ascontiguousarray(x, dtype=np.float64) for x in w.get("gv", [eq] * self.v0.nstream)]
        self._h = (C.c_void_p * nv)(*[v.h for v in self.voices])

    def _weights(self):
        w = Weights()
        w.duration = _ptr(self.w_duration, C.c_double)
        for i in range(self.v0.nstream):
            w.parameter[i] = _ptr(self.w_parameter[i], C.c_double)
            w.gv[i] = _ptr(self.w_gv[i], C.c_double)
        return w

    def duration_params(self, labels):
        """Models::duration() over the set (src/model/mod.rs:80-92): [S][2] blended (mean, variance)."""
        out = np.zeros((len(labels) * self.v0.nstate, 2))
        if self.L.jbo_duration_params_multi(self._h, len(self.voices), _ptr(self.w_duration, C.c_double), _strs(labels),
                                            len(labels), out.ctypes.data):
            raise RuntimeError("duration_params_multi")
        return out

    def durations(self, labels, speed=1.0, times=None):
        out = np.zeros(len(labels) * self.v0.nstate, dtype=np.uint32)
        t = None if times is None else np.ascontiguousarray(times, dtype=np.float64)
        if self.L.jbo_durations_multi(self._h, len(self.voices), _ptr(self.w_duration, C.c_double), _strs(labels),
                                      len(labels), float(speed), t.ctypes.data if t is not None else None,
                                      out.ctypes.data):
            raise RuntimeError("durations_multi")
        return out

    def stream_states(self, stream, labels, gv_weight=1.0, msd_threshold=0.5) -> StreamStates:
        v0 = self.v0
        S = len(labels) * v0.nstate
        L, W = v0.vector_length[stream], v0.num_windows[stream]
        mean, var, msd = np.zeros((S, W * L)), np.zeros((S, W * L)), np.zeros(S)
        nv = len(self.voices)
        if self.L.jbo_stream_params_multi(self._h, nv, _ptr(self.w_parameter[stream], C.c_double), stream,
                                          _strs(labels), len(labels), mean.ctypes.data, var.ctypes.data,
                                          msd.ctypes.data):
            raise RuntimeError("stream_params_multi")
        gm = gv = gs = None
        if v0.use_gv[stream] and len(labels):
            gm, gv, gs = np.zeros(L), np.zeros(L), np.zeros(S, dtype=np.uint8)
            if self.L.jbo_gv_params_multi(self._h, nv, _ptr(self.w_gv[stream], C.c_double), stream, _strs(labels),
                                          len(labels), gm.ctypes.data, gv.ctypes.data, gs.ctypes.data):
                raise RuntimeError("gv_params_multi")
        ww = [len(w) for w in v0.windows[stream]]
        wc = [c for w in v0.windows[stream] for c in w]
        return StreamStates(L, W, v0.is_msd[stream], v0.use_gv[stream], ww, wc, mean, var, msd, gm, gv, gs,
                            gv_weight, msd_threshold)

    def synthesize(self, lines, speed=1.0, volume=1.0, half_tone=0.0, alignment=False, beta=0.0,
                   want_tracks=False):
        c = Cond()
        self.L.jbo_cond_default(C.byref(c))
        c.speed, c.volume, c.additional_half_tone, c.beta = speed, volume, half_tone, beta
        c.phoneme_alignment = int(alignment)
        w = self._weights()
        pcm, n = C.POINTER(C.c_double)(), C.c_size_t()
        dur, S = C.POINTER(C.c_uint32)(), C.c_uint32()
        mcp, lf0, lpf, T = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.c_size_t()
        r = self.L.jbo_synthesize_multi_ex(self._h, len(self.voices), C.byref(w), C.byref(c), _strs(lines),
                                           len(lines), C.byref(pcm), C.byref(n), C.byref(dur), C.byref(S),
                                           C.byref(mcp), C.byref(lf0), C.byref(lpf), C.byref(T))
        if r:
            raise RuntimeError(f"oracle synthesize_multi failed: {r}")

        def take(p, shape):
            if not p or int(np.prod(shape)) == 0:
                return np.zeros(shape)
            return np.ctypeslib.as_array(p, shape=shape).copy()

        out = take(pcm, (n.value,))
        res = out
        if want_tracks:
            Tn, v0 = T.value, self.v0
            res = dict(pcm=out, T=Tn,
                       dur=(np.ctypeslib.as_array(dur, shape=(S.value,)).copy() if dur else np.zeros(0, np.uint32)),
                       mcp=take(mcp, (Tn, v0.vector_length[0])), lf0=take(lf0, (Tn,)),
                       lpf=take(lpf, (Tn, v0.vector_length[2] if v0.nstream > 2 else 0)))
        for p in (pcm, dur, mcp, lf0, lpf):
            if p:
                self.L.jbo_free(p)
        return res


def mask(st: StreamStates, dur):
    dur = np.ascontiguousarray(dur, dtype=np.uint32)
    T = int(dur.sum())
    m = np.zeros(max(T, 1), dtype=np.uint8)
    s = st.c_struct()
    lib().jbo_mask(C.byref(s), len(dur), dur.ctypes.data, m.ctypes.data)
    return m[:T]


def boundary_distances(m):
    m = np.ascontiguousarray(m, dtype=np.uint8)
    T = len(m)
    l = np.zeros(max(T, 1), dtype=np.uint64)
    r = np.zeros(max(T, 1), dtype=np.uint64)
    lib().jbo_boundary_distances(m.ctypes.data, T, l.ctypes.data, r.ctypes.data)
    return l[:T], r[:T]


def mlpg(st: StreamStates, dur):
    dur = np.ascontiguousarray(dur, dtype=np.uint32)
    T = int(dur.sum())
    par = np.zeros((max(T, 1), st.L))
    s = st.c_struct()
    if lib().jbo_mlpg(C.byref(s), len(dur), dur.ctypes.data, par.ctypes.data):
        raise RuntimeError("mlpg")
    return par[:T]


def freqt(c1, m2, alpha):
    """freqt (cepstrum.rs:153-173) in the reference's (ascending) input order."""
    c1 = np.ascontiguousarray(c1, dtype=np.float64)
    out = np.zeros(m2 + 1)
    lib().jbo_freqt(c1.ctypes.data, len(c1), out.ctypes.data, m2, alpha)
    return out


def c2ir(c, n):
    c = np.ascontiguousarray(c, dtype=np.float64)
    ir = np.zeros(n)
    lib().jbo_c2ir(c.ctypes.data, len(c), ir.ctypes.data, n)
    return ir


def b2en(b, alpha):
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().jbo_b2en(b.ctypes.data, len(b), alpha)


def postfilter_mcp(mc, alpha, beta):
    """MelCepstrum::postfilter_mcp (cepstrum.rs:23-37); returns the filtered copy."""
    out = np.array(mc, dtype=np.float64, copy=True)
    lib().jbo_postfilter_mcp(out.ctypes.data, len(out), alpha, beta)
    return out


def vocoder(fs, fperiod, alpha, volume, lf0, mcp, lpf, dumps=False, beta=0.0, stage=0, use_log_gain=False,
            coef=None, cfirst=None):
    """Vocoder::synthesize over all frames.  stage > 0: the Stage::NonZero branch (spectrum = [gain, LSP...],
    MGLSA filter; vocoder/mod.rs:90-107,142-176) -- parity unpinned."""
    lf0 = np.ascontiguousarray(lf0, dtype=np.float64).reshape(-1)
    T = len(lf0)
    mcp = np.ascontiguousarray(mcp, dtype=np.float64).reshape(T, -1)
    nmcp = mcp.shape[1]
    nlpf = 0
    if lpf is not None and np.size(lpf):
        lpf = np.ascontiguousarray(lpf, dtype=np.float64).reshape(T, -1)
        nlpf = lpf.shape[1]
    pcm = np.zeros(T * fperiod)
    exc = np.zeros(T * fperiod) if dumps else None
    pul = np.zeros(T * fperiod) if dumps else None
    if stage:
        if coef is not None:
            coef = np.ascontiguousarray(coef, dtype=np.float64).reshape(T, nmcp)
            cfirst = np.ascontiguousarray(cfirst, dtype=np.float64).reshape(nmcp)
        r = lib().jbo_vocoder_stage_coef(fs, fperiod, alpha, beta, volume, int(stage), int(bool(use_log_gain)), nmcp,
                                         nlpf, T, lf0.ctypes.data, mcp.ctypes.data, lpf.ctypes.data if nlpf else None,
                                         coef.ctypes.data if coef is not None else None,
                                         cfirst.ctypes.data if coef is not None else None,
                                         pcm.ctypes.data, exc.ctypes.data if dumps else None)
        if r:
            raise RuntimeError("vocoder_stage")
        return (pcm, exc, None) if dumps else pcm
    r = lib().jbo_vocoder_beta(fs, fperiod, alpha, beta, volume, nmcp, nlpf, T, lf0.ctypes.data, mcp.ctypes.data,
                          lpf.ctypes.data if nlpf else None, pcm.ctypes.data,
                          exc.ctypes.data if dumps else None, pul.ctypes.data if dumps else None)
    if r:
        raise RuntimeError("vocoder")
    return (pcm, exc, pul) if dumps else pcm


# ---- X2 pieces (Stage::NonZero; parity unpinned, held by tests/test_oracle_stage.py) ----
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def lsp2lpc(lsp):
    lsp = _f64(lsp)
    out = np.zeros(len(lsp) + 1)
    lib().jbo_lsp2lpc(lsp.ctypes.data, len(lsp), out.ctypes.data)
    return out


def gnorm(c, gamma):
    c = _f64(c).copy()
    lib().jbo_gnorm(c.ctypes.data, len(c), gamma)
    return c


def ignorm(c, gamma):
    c = _f64(c).copy()
    lib().jbo_ignorm(c.ctypes.data, len(c), gamma)
    return c


def gc2gc(c1, g1, m2, g2):
    c1 = _f64(c1)
    out = np.zeros(m2 + 1)
    lib().jbo_gc2gc(c1.ctypes.data, len(c1), g1, out.ctypes.data, m2, g2)
    return out


def mgc2mgc(c1, a1, g1, m2, a2, g2):
    c1 = _f64(c1)
    out = np.zeros(m2 + 1)
    lib().jbo_mgc2mgc(c1.ctypes.data, len(c1), a1, g1, out.ctypes.data, m2, a2, g2)
    return out


def lsp2mgc(lsp, alpha, use_log_gain, stage):
    lsp = _f64(lsp)
    out = np.zeros(len(lsp))
    lib().jbo_lsp2mgc(lsp.ctypes.data, len(lsp), alpha, int(bool(use_log_gain)), stage, -1.0 / stage, out.ctypes.data)
    return out


def postfilter_lsp(lsp, alpha, use_log_gain, stage, beta):
    lsp = _f64(lsp).copy()
    lib().jbo_postfilter_lsp(lsp.ctypes.data, len(lsp), alpha, int(bool(use_log_gain)), stage, -1.0 / stage, beta)
    return lsp


def check_lsp_stability(lsp):
    lsp = _f64(lsp).copy()
    lib().jbo_check_lsp_stability(lsp.ctypes.data, len(lsp))
    return lsp


def stage_coefficients(spectrum, alpha, beta, use_log_gain, stage, filtered=True):
    sp = _f64(spectrum)
    out = np.zeros(len(sp))
    lib().jbo_stage_coefficients(sp.ctypes.data, len(sp), alpha, beta, int(bool(use_log_gain)), stage,
                                 int(bool(filtered)), out.ctypes.data)
    return out


def mglsa_df(d, x, alpha, c):
    """one sample through the MGLSA cascade; d [stage, n] is updated in place; returns the output"""
    assert d.dtype == np.float64 and d.flags["C_CONTIGUOUS"]
    c = _f64(c)
    xv = np.array([x], dtype=np.float64)
    lib().jbo_mglsa_df(d.ctypes.data, d.shape[0], d.shape[1], xv.ctypes.data, alpha, c.ctypes.data)
    return float(xv[0])


def noise(n):
    out = np.zeros(n)
    lib().jbo_noise(out.ctypes.data, n)
    return out
