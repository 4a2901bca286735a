"""Host-side mirror of jbonsai's public API over the C ABI.

  Engine.load / load_from_bytes / synthesize / generator   src/engine.rs:257-366
  Condition setters/getters                                src/engine.rs:127-243
  SpeechGenerator.{fperiod, synthesized_frames, generate_step, generate_all}
                                                           src/speech.rs:53-96
Names, argument meaning and error behaviour follow the reference so that the
tests read like the reference's own (src/lib.rs:38-160).
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Sequence

import numpy as np

from . import _ffi as F
from .batch import StreamInfo, StreamStates, Utterance, VoiceInfo


def _bind(L):
    if getattr(L, "_engine_bound", False):
        return
    vp, sz, dp = C.c_void_p, C.c_size_t, C.POINTER(C.c_double)
    cpp = C.POINTER(C.c_char_p)
    L.jb_engine_load.argtypes = [cpp, sz, C.POINTER(vp)]
    L.jb_engine_load_from_bytes.argtypes = [C.POINTER(C.c_char_p), C.POINTER(sz), sz, C.POINTER(vp)]
    L.jb_engine_free.argtypes = [vp]
    L.jb_engine_free.restype = None
    for n in ("sampling_frequency", "fperiod"):
        getattr(L, "jb_engine_set_" + n).argtypes = [vp, sz]
        getattr(L, "jb_engine_get_" + n).argtypes = [vp]
        getattr(L, "jb_engine_get_" + n).restype = sz
    for n in ("volume", "speed", "alpha", "beta", "additional_half_tone"):
        getattr(L, "jb_engine_set_" + n).argtypes = [vp, C.c_double]
        getattr(L, "jb_engine_get_" + n).argtypes = [vp]
        getattr(L, "jb_engine_get_" + n).restype = C.c_double
    for n in ("msd_threshold", "gv_weight"):
        getattr(L, "jb_engine_set_" + n).argtypes = [vp, sz, C.c_double]
        getattr(L, "jb_engine_get_" + n).argtypes = [vp, sz]
        getattr(L, "jb_engine_get_" + n).restype = C.c_double
    L.jb_engine_set_phoneme_alignment_flag.argtypes = [vp, C.c_int]
    L.jb_engine_get_phoneme_alignment_flag.argtypes = [vp]
    L.jb_engine_set_batch_invariant.argtypes = [vp, C.c_int]
    L.jb_engine_get_batch_invariant.argtypes = [vp]
    for n in ("num_voices", "num_streams", "num_states"):
        getattr(L, "jb_engine_" + n).argtypes = [vp]
        getattr(L, "jb_engine_" + n).restype = sz
    L.jb_engine_set_interpolation_weight.argtypes = [vp, C.c_int, sz, dp, sz]
    L.jb_engine_get_interpolation_weight.argtypes = [vp, C.c_int, sz, dp, sz, C.POINTER(sz)]
    L.jb_engine_new.argtypes = [vp, vp, C.POINTER(vp)]
    L.jb_engine_model_shape.argtypes = [vp, sz, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.jb_engine_pdf_table.argtypes = [vp, sz, C.c_int, sz, C.POINTER(C.POINTER(C.c_float)), C.POINTER(sz)]
    L.jb_engine_tree_index.argtypes = [vp, sz, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]
    L.jb_synthesize.argtypes = [vp, cpp, sz, C.POINTER(dp), C.POINTER(sz)]
    L.jb_pcm_free.argtypes = [dp]
    L.jb_pcm_free.restype = None
    L.jb_synthesize_batch.argtypes = [vp, cpp, C.POINTER(sz), sz, C.c_int32, C.POINTER(dp), C.POINTER(sz)]
    L.jb_synthesize_batch_i16.argtypes = [vp, cpp, C.POINTER(sz), sz, C.c_int32,
                                          C.POINTER(C.POINTER(C.c_int16)), C.POINTER(sz)]
    L.jb_synthesize_batch_multi.argtypes = [vp, cpp, C.POINTER(sz), sz, C.POINTER(C.c_int32), sz, C.POINTER(dp),
                                            C.POINTER(sz)]
    L.jb_synthesize_batch_i16_multi.argtypes = [vp, cpp, C.POINTER(sz), sz, C.POINTER(C.c_int32), sz,
                                                C.POINTER(C.POINTER(C.c_int16)), C.POINTER(sz)]
    L.jb_pcm_i16_free.argtypes = [C.POINTER(C.c_int16)]
    L.jb_pcm_i16_free.restype = None
    L.jb_engine_states.argtypes = [vp, cpp, sz, C.POINTER(vp)]
    L.jb_states_utt.argtypes = [vp]
    L.jb_states_utt.restype = C.POINTER(F.StateUtt)
    L.jb_engine_voice_desc.argtypes = [vp]
    L.jb_engine_voice_desc.restype = C.POINTER(F.VoiceDesc)
    L.jb_states_duration_params.argtypes = [vp]
    L.jb_states_duration_params.restype = C.POINTER(C.c_double)
    L.jb_states_free.argtypes = [vp]
    L.jb_states_free.restype = None
    L.jb_generator_new.argtypes = [vp, cpp, sz, C.POINTER(vp)]
    for n in ("fperiod", "synthesized_frames", "total_frames"):
        getattr(L, "jb_generator_" + n).argtypes = [vp]
        getattr(L, "jb_generator_" + n).restype = sz
    L.jb_generator_step.argtypes = [vp, dp, sz]
    L.jb_generator_step.restype = C.c_long
    L.jb_generator_step_n.argtypes = [vp, dp, sz, sz]
    L.jb_generator_step_n.restype = C.c_long
    L.jb_generator_free.argtypes = [vp]
    L.jb_generator_free.restype = None
    L._engine_bound = True


def _lines(labels: Sequence[str]):
    arr = (C.c_char_p * max(1, len(labels)))()
    for i, s in enumerate(labels):
        arr[i] = s.encode() if isinstance(s, str) else bytes(s)
    return arr


class _Condition:
    """View of the engine's Condition (src/engine.rs:31-243)."""

    def __init__(self, eng):
        self._e = eng

    def _h(self):
        return self._e._h

    def _L(self):
        return self._e._L

    def set_sampling_frequency(self, i): F.check(self._L().jb_engine_set_sampling_frequency(self._h(), int(i)))
    def get_sampling_frequency(self): return self._L().jb_engine_get_sampling_frequency(self._h())
    def set_fperiod(self, i): F.check(self._L().jb_engine_set_fperiod(self._h(), int(i)))
    def get_fperiod(self): return self._L().jb_engine_get_fperiod(self._h())
    def set_volume(self, db): F.check(self._L().jb_engine_set_volume(self._h(), float(db)))
    def get_volume(self): return self._L().jb_engine_get_volume(self._h())
    def set_msd_threshold(self, s, f): F.check(self._L().jb_engine_set_msd_threshold(self._h(), s, float(f)))
    def get_msd_threshold(self, s): return self._L().jb_engine_get_msd_threshold(self._h(), s)
    def set_gv_weight(self, s, f): F.check(self._L().jb_engine_set_gv_weight(self._h(), s, float(f)))
    def get_gv_weight(self, s): return self._L().jb_engine_get_gv_weight(self._h(), s)
    def set_speed(self, f): F.check(self._L().jb_engine_set_speed(self._h(), float(f)))
    def get_speed(self): return self._L().jb_engine_get_speed(self._h())
    def set_phoneme_alignment_flag(self, b): F.check(self._L().jb_engine_set_phoneme_alignment_flag(self._h(), int(bool(b))))
    def get_phoneme_alignment_flag(self): return bool(self._L().jb_engine_get_phoneme_alignment_flag(self._h()))
    def set_batch_invariant(self, b): F.check(self._L().jb_engine_set_batch_invariant(self._h(), int(bool(b))))
    def get_batch_invariant(self): return bool(self._L().jb_engine_get_batch_invariant(self._h()))
    def set_alpha(self, f): F.check(self._L().jb_engine_set_alpha(self._h(), float(f)))
    def get_alpha(self): return self._L().jb_engine_get_alpha(self._h())
    def set_beta(self, f): F.check(self._L().jb_engine_set_beta(self._h(), float(f)))
    def get_beta(self): return self._L().jb_engine_get_beta(self._h())
    def set_additional_half_tone(self, f): F.check(self._L().jb_engine_set_additional_half_tone(self._h(), float(f)))
    def get_additional_half_tone(self): return self._L().jb_engine_get_additional_half_tone(self._h())

    # InterporationWeight (src/model/interporation_weight.rs:48-126)
    def _setw(self, which, stream, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        F.check(self._L().jb_engine_set_interpolation_weight(
            self._h(), which, stream, a.ctypes.data_as(C.POINTER(C.c_double)), len(a)))

    def _getw(self, which, stream):
        n = C.c_size_t()
        F.check(self._L().jb_engine_get_interpolation_weight(self._h(), which, stream, None, 0, C.byref(n)))
        a = np.zeros(n.value)
        F.check(self._L().jb_engine_get_interpolation_weight(
            self._h(), which, stream, a.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        return a

    def get_interpolation_duration(self): return self._getw(0, 0)
    def get_interpolation_parameter(self, stream): return self._getw(1, stream)
    def get_interpolation_gv(self, stream): return self._getw(2, stream)
    def set_interpolation_duration(self, w): self._setw(0, 0, w)
    def set_interpolation_parameter(self, stream, w): self._setw(1, stream, w)
    def set_interpolation_gv(self, stream, w): self._setw(2, stream, w)


class Engine:
    """jbonsai::Engine over libjbonsai_amd.so."""

    def __init__(self, handle, L):
        self._h, self._L = handle, L
        self.condition = _Condition(self)

    @classmethod
    def load(cls, voices: Sequence[str]) -> "Engine":
        L = F.lib()
        _bind(L)
        h = C.c_void_p()
        paths = [str(p) for p in voices]
        F.check(L.jb_engine_load(_lines(paths), len(paths), C.byref(h)))
        return cls(h, L)

    @classmethod
    def load_from_bytes(cls, voices: Sequence[bytes]) -> "Engine":
        L = F.lib()
        _bind(L)
        h = C.c_void_p()
        bufs = (C.c_char_p * max(1, len(voices)))(*[C.c_char_p(b) for b in voices])
        lens = (C.c_size_t * max(1, len(voices)))(*[len(b) for b in voices])
        F.check(L.jb_engine_load_from_bytes(bufs, lens, len(voices), C.byref(h)))
        return cls(h, L)

    @classmethod
    def new(cls, voices_of: "Engine", condition_of: "Engine") -> "Engine":
        """Engine::new(voices, condition) (src/engine.rs:289-291): the voices of one engine (shared) with a
        copy of the Condition of another; Engine.new(e, e) is Engine::clone."""
        h = C.c_void_p()
        F.check(voices_of._L.jb_engine_new(voices_of._h, condition_of._h, C.byref(h)))
        return cls(h, voices_of._L)

    def clone(self) -> "Engine":
        return Engine.new(self, self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_engine_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection --
    @property
    def num_voices(self): return self._L.jb_engine_num_voices(self._h)
    @property
    def num_streams(self): return self._L.jb_engine_num_streams(self._h)
    @property
    def num_states(self): return self._L.jb_engine_num_states(self._h)

    def model_shape(self, kind, voice=0):
        nt, pl = C.c_size_t(), C.c_size_t()
        F.check(self._L.jb_engine_model_shape(self._h, voice, kind, C.byref(nt), C.byref(pl)))
        return nt.value, pl.value

    def pdf_table(self, kind, tree, voice=0) -> np.ndarray:
        _, pl = self.model_shape(kind, voice)
        p, n = C.POINTER(C.c_float)(), C.c_size_t()
        F.check(self._L.jb_engine_pdf_table(self._h, voice, kind, tree, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(p, shape=(n.value, pl)).copy()

    def tree_index(self, kind, state_index, label, voice=0):
        ts, pi = C.c_int(), C.c_int()
        F.check(self._L.jb_engine_tree_index(self._h, voice, kind, state_index, label.encode(),
                                             C.byref(ts), C.byref(pi)))
        return (None if ts.value < 0 else ts.value, pi.value)

    def voice_info(self) -> VoiceInfo:
        d = self._L.jb_engine_voice_desc(self._h).contents
        streams = []
        for i in range(d.nstream):
            s = d.stream[i]
            off, wins = 0, []
            for w in range(s.num_windows):
                n = s.win_width[w]
                wins.append([s.win_coef[off + k] for k in range(n)])
                off += n
            streams.append(StreamInfo(s.vector_length, bool(s.is_msd), bool(s.use_gv), wins))
        return VoiceInfo(d.sampling_frequency, d.fperiod, d.alpha, streams, volume=d.volume,
                         beta=d.beta, stage=d.stage, use_log_gain=bool(d.use_log_gain))

    def states(self, labels: Sequence[str]) -> Utterance:
        """Front half only: labels -> state-level utterance (durations + per-state pdfs)."""
        h = C.c_void_p()
        F.check(self._L.jb_engine_states(self._h, _lines(labels), len(labels), C.byref(h)))
        try:
            u = self._L.jb_states_utt(h).contents
            vi = self.voice_info()
            S = u.num_states
            dur = np.ctypeslib.as_array(u.durations, shape=(S,)).copy() if S else np.zeros(0, np.uint32)
            sts = []
            for i, si in enumerate(vi.streams):
                s = u.stream[i]
                WL = si.vector_length * len(si.windows)
                arr = lambda p, shape, t=np.float64: (np.ctypeslib.as_array(p, shape=shape).copy()
                                                      if p and int(np.prod(shape)) else
                                                      (np.zeros(shape, t) if p or shape[0] == 0 else None))
                mean, var = arr(s.mean, (S, WL)), arr(s.var, (S, WL))
                if mean is None:
                    mean, var = np.zeros((S, WL)), np.zeros((S, WL))
                msd = arr(s.msd, (S,)) if s.msd else None
                gm = arr(s.gv_mean, (si.vector_length,)) if s.gv_mean else None
                gv = arr(s.gv_var, (si.vector_length,)) if s.gv_var else None
                gs = (np.ctypeslib.as_array(s.gv_switch, shape=(S,)).copy() if s.gv_switch and S else None)
                sts.append(StreamStates(mean, var, msd, gm, gv, gs, s.gv_weight, s.msd_threshold))
            return Utterance(dur, sts)
        finally:
            self._L.jb_states_free(h)

    def duration_params(self, labels: Sequence[str]) -> np.ndarray:
        """Models::duration() (src/model/mod.rs:80-92): [S][2] blended (mean, variance) of the duration pdfs."""
        h = C.c_void_p()
        F.check(self._L.jb_engine_states(self._h, _lines(labels), len(labels), C.byref(h)))
        try:
            S = self._L.jb_states_utt(h).contents.num_states
            p = self._L.jb_states_duration_params(h)
            return np.ctypeslib.as_array(p, shape=(S, 2)).copy() if S and p else np.zeros((0, 2))
        finally:
            self._L.jb_states_free(h)

    # -- synthesis --
    def synthesize(self, labels: Sequence[str]) -> np.ndarray:
        pcm, n = C.POINTER(C.c_double)(), C.c_size_t()
        F.check(self._L.jb_synthesize(self._h, _lines(labels), len(labels), C.byref(pcm), C.byref(n)))
        try:
            return np.ctypeslib.as_array(pcm, shape=(n.value,)).copy() if n.value else np.zeros(0)
        finally:
            if pcm:
                self._L.jb_pcm_free(pcm)

    def synthesize_batch(self, utterances: Sequence[Sequence[str]], device: int = -1,
                         i16: bool = False, devices: Sequence[int] = None) -> List[np.ndarray]:
        """jb_synthesize_batch / jb_synthesize_batch_i16 (or, with `devices`, their _multi forms: the
        utterances are LPT-split over the listed GPUs, one host thread per device).  The arrays view
        the library-owned buffers (no copy); each is released with jb_pcm_free / jb_pcm_i16_free when
        its array dies."""
        flat = [l for u in utterances for l in u]
        off = np.cumsum([0] + [len(u) for u in utterances]).astype(np.uint64)
        B = len(utterances)
        offs = (C.c_size_t * (B + 1))(*[int(x) for x in off])
        ety = C.c_int16 if i16 else C.c_double
        pcm = (C.POINTER(ety) * max(1, B))()
        ns = (C.c_size_t * max(1, B))()
        free = self._L.jb_pcm_i16_free if i16 else self._L.jb_pcm_free
        if devices is not None:
            fn = self._L.jb_synthesize_batch_i16_multi if i16 else self._L.jb_synthesize_batch_multi
            dv = (C.c_int32 * max(1, len(devices)))(*[int(d) for d in devices])
            F.check(fn(self._h, _lines(flat), offs, B, dv, len(devices), pcm, ns))
        else:
            fn = self._L.jb_synthesize_batch_i16 if i16 else self._L.jb_synthesize_batch
            F.check(fn(self._h, _lines(flat), offs, B, device, pcm, ns))
        out = []
        for i in range(B):
            if not ns[i]:
                out.append(np.zeros(0, dtype=np.int16 if i16 else np.float64))
                continue
            buf = (ety * ns[i]).from_address(C.addressof(pcm[i].contents))
            arr = np.frombuffer(buf, dtype=np.int16 if i16 else np.float64)
            weakref.finalize(buf, free, C.cast(C.addressof(pcm[i].contents), C.POINTER(ety)))
            out.append(arr)
        return out

    def generator(self, labels: Sequence[str]) -> "SpeechGenerator":
        h = C.c_void_p()
        F.check(self._L.jb_generator_new(self._h, _lines(labels), len(labels), C.byref(h)))
        return SpeechGenerator(h, self._L)


class SpeechGenerator:
    """jbonsai::speech::SpeechGenerator (src/speech.rs:9-96)."""

    def __init__(self, handle, L):
        self._h, self._L = handle, L

    def fperiod(self): return self._L.jb_generator_fperiod(self._h)
    def synthesized_frames(self): return self._L.jb_generator_synthesized_frames(self._h)
    def total_frames(self): return self._L.jb_generator_total_frames(self._h)

    def generate_step(self, speech: np.ndarray) -> int:
        """Writes fperiod samples to speech[0:fperiod]; returns fperiod, or 0 when exhausted."""
        assert speech.dtype == np.float64 and speech.flags["C_CONTIGUOUS"]
        r = self._L.jb_generator_step(self._h, speech.ctypes.data_as(C.POINTER(C.c_double)), speech.size)
        if r < 0:
            F.check(int(r))
        return int(r)

    def generate_steps(self, speech: np.ndarray, max_frames: int) -> int:
        """Up to max_frames generate_step calls in one (jb_generator_step_n): returns the samples written."""
        assert speech.dtype == np.float64 and speech.flags["C_CONTIGUOUS"]
        r = self._L.jb_generator_step_n(self._h, speech.ctypes.data_as(C.POINTER(C.c_double)), speech.size,
                                        int(max_frames))
        if r < 0:
            F.check(int(r))
        return int(r)

    def generate_all(self) -> np.ndarray:
        """generate_all (src/speech.rs:87-96): the frames not yet synthesized."""
        fp = self.fperiod()
        left = self.total_frames() - self.synthesized_frames()
        buf = np.zeros(left * fp)
        if left:
            got = self.generate_steps(buf, left)
            assert got == left * fp
        return buf

    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_generator_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
