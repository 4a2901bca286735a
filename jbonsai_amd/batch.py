"""State-level batch API: the image of ModelStream + durations for B utterances
(reference: src/model/model_stream.rs:6-15, src/engine.rs:321-365)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi as F


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


@dataclass
class StreamInfo:
    """Static stream description (vector length, MSD/GV flags, delta windows)."""
    vector_length: int
    is_msd: bool
    use_gv: bool
    windows: List[List[float]]


@dataclass
class VoiceInfo:
    sampling_frequency: int
    fperiod: int
    alpha: float
    streams: List[StreamInfo]
    volume: float = 1.0
    beta: float = 0.0
    stage: int = 0
    use_log_gain: bool = False

    def c_struct(self):
        v = F.VoiceDesc()
        v.sampling_frequency, v.fperiod = self.sampling_frequency, self.fperiod
        v.nstream, v.stage, v.use_log_gain = len(self.streams), self.stage, int(self.use_log_gain)
        v.alpha, v.beta, v.volume = self.alpha, self.beta, self.volume
        keep = []
        for i, s in enumerate(self.streams):
            d = v.stream[i]
            d.vector_length, d.num_windows = s.vector_length, len(s.windows)
            d.is_msd, d.use_gv = int(s.is_msd), int(s.use_gv)
            coef = np.array([c for w in s.windows for c in w], dtype=np.float64)
            for k, w in enumerate(s.windows):
                d.win_width[k] = len(w)
            d.win_coef = _dp(coef)
            keep.append(coef)
        return v, keep


@dataclass
class StreamStates:
    mean: np.ndarray            # [S, W*L]
    var: np.ndarray             # [S, W*L]
    msd: Optional[np.ndarray] = None      # [S]
    gv_mean: Optional[np.ndarray] = None  # [L]
    gv_var: Optional[np.ndarray] = None
    gv_switch: Optional[np.ndarray] = None  # [S] uint8
    gv_weight: float = 1.0
    msd_threshold: float = 0.5

    def __post_init__(self):
        f = lambda a, t: None if a is None else np.ascontiguousarray(a, dtype=t)
        self.mean, self.var = f(self.mean, np.float64), f(self.var, np.float64)
        self.msd, self.gv_mean, self.gv_var = f(self.msd, np.float64), f(self.gv_mean, np.float64), f(self.gv_var, np.float64)
        self.gv_switch = f(self.gv_switch, np.uint8)


@dataclass
class Utterance:
    durations: np.ndarray       # [S] uint32
    streams: List[StreamStates] = field(default_factory=list)

    def __post_init__(self):
        self.durations = np.ascontiguousarray(self.durations, dtype=np.uint32)

    def c_struct(self):
        u = F.StateUtt()
        u.num_states = len(self.durations)
        u.durations = self.durations.ctypes.data_as(C.POINTER(C.c_uint32))
        for i, s in enumerate(self.streams):
            d = u.stream[i]
            d.mean, d.var, d.msd = _dp(s.mean), _dp(s.var), _dp(s.msd)
            d.gv_mean, d.gv_var = _dp(s.gv_mean), _dp(s.gv_var)
            d.gv_switch = s.gv_switch.ctypes.data_as(C.POINTER(C.c_uint8)) if s.gv_switch is not None else None
            d.gv_weight, d.msd_threshold = s.gv_weight, s.msd_threshold
        return u


class PdfSet:
    """Device-resident pdf tables of a voice set (jb_pdf_set_create): tables[v][s] = float32 array
    [n_rows, row_len] of stream s of voice v, all trees of the stream concatenated."""

    def __init__(self, tables: Sequence[Sequence[np.ndarray]], device: int = -1):
        L = F.lib()
        self._L = L
        nv, ns = len(tables), len(tables[0])
        self._keep = [[np.ascontiguousarray(t, dtype=np.float32) for t in row] for row in tables]
        arr = (F.PdfTable * (nv * ns))()
        for v in range(nv):
            for s in range(ns):
                t = self._keep[v][s]
                arr[v * ns + s].rows = t.ctypes.data_as(C.POINTER(C.c_float))
                arr[v * ns + s].n_rows, arr[v * ns + s].row_len = t.shape
        h = C.c_void_p()
        F.check(L.jb_pdf_set_create(arr, nv, ns, device, C.byref(h)))
        self._h, self.n_voices, self.nstream, self.device = h, nv, ns, device

    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_pdf_set_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class IndexStreamStates:
    """One stream of an utterance as pdf row indices (jb_index_stream): rows[v] = uint32 [S] rows of
    voice v's table; weights [n_voices]."""
    rows: List[np.ndarray]
    weights: np.ndarray
    gv_mean: Optional[np.ndarray] = None
    gv_var: Optional[np.ndarray] = None
    gv_switch: Optional[np.ndarray] = None
    gv_weight: float = 1.0
    msd_threshold: float = 0.5

    def __post_init__(self):
        f = lambda a, t: None if a is None else np.ascontiguousarray(a, dtype=t)
        self.rows = [np.ascontiguousarray(r, dtype=np.uint32) for r in self.rows]
        self.weights = np.ascontiguousarray(self.weights, dtype=np.float64)
        self.gv_mean, self.gv_var = f(self.gv_mean, np.float64), f(self.gv_var, np.float64)
        self.gv_switch = f(self.gv_switch, np.uint8)

    def __setattr__(self, name, value):
        # (an IndexUtterance that has marshalled this stream keys its cached struct on the identity of every array it
        # points into: any replacement, of a field or of an element of `rows`, makes it marshal again)
        object.__setattr__(self, name, value)


@dataclass
class IndexUtterance:
    durations: np.ndarray
    streams: List[IndexStreamStates] = field(default_factory=list)
    lf0_offset: float = 0.0

    def __post_init__(self):
        self.durations = np.ascontiguousarray(self.durations, dtype=np.uint32)

    def c_struct(self):
        # marshalled once per object (the struct only points into arrays this object owns; an utterance is not
        # edited after its first use): a job that creates its batches again every pass -- bench.py's config-3
        # job -- spent 12 of the 27 ms of a 512-utterance creation building these
        # -- until one is.  The cached struct is valid for exactly the arrays it points into: the key is the identity
        # of every array (and the value of every scalar) placed in it, and the arrays are kept referenced beside it,
        # so that replacing an element in place (u.streams[i] = other, st.rows[v] = new) can neither leave the struct
        # pointing into freed memory nor go unseen (a sum of version counters missed both)
        arrays = [self.durations]
        for s in self.streams:
            arrays += list(s.rows) + [s.weights, s.gv_mean, s.gv_var, s.gv_switch]
        key = (tuple(id(a) for a in arrays), self.lf0_offset,
               tuple((s.gv_weight, s.msd_threshold) for s in self.streams))
        c = self.__dict__.get("_c")
        if c is not None and self.__dict__.get("_c_key") == key:
            return c
        u = F.IndexUtt()
        u.num_states = len(self.durations)
        u.durations = self.durations.ctypes.data_as(C.POINTER(C.c_uint32))
        u.lf0_offset = self.lf0_offset
        for i, s in enumerate(self.streams):
            d = u.stream[i]
            for v, r in enumerate(s.rows):
                d.row[v] = r.ctypes.data_as(C.POINTER(C.c_uint32))
            d.weight = _dp(s.weights)
            d.gv_mean, d.gv_var = _dp(s.gv_mean), _dp(s.gv_var)
            d.gv_switch = s.gv_switch.ctypes.data_as(C.POINTER(C.c_uint8)) if s.gv_switch is not None else None
            d.gv_weight, d.msd_threshold = s.gv_weight, s.msd_threshold
        self.__dict__["_c"] = u
        self.__dict__["_c_key"] = key
        self.__dict__["_c_arrays"] = arrays  # what the struct points into stays alive as long as the struct
        return u

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        self.__dict__.pop("_c", None)


@dataclass
class TrackUtterance:
    """The three parameter tracks of one utterance as SpeechGenerator::new takes them (src/speech.rs:25-50):
    spectrum [T][nmcp], lf0 [T][1] (NODATA = unvoiced frame), lpf [T][nlpf]."""
    spectrum: np.ndarray
    lf0: np.ndarray
    lpf: np.ndarray

    def __post_init__(self):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.spectrum, self.lpf = f(self.spectrum), f(self.lpf)
        self.lf0 = f(self.lf0)
        if self.lf0.ndim == 1:
            self.lf0 = self.lf0.reshape(-1, 1)

    def c_struct(self):
        u = F.TrackUtt()
        u.n_spectrum, u.n_lf0, u.n_lpf = len(self.spectrum), len(self.lf0), len(self.lpf)
        w = lambda a: int(a.shape[1]) if a.ndim == 2 else 0
        u.spectrum_width, u.lf0_width, u.lpf_width = w(self.spectrum), w(self.lf0), w(self.lpf)
        u.spectrum, u.lf0, u.lpf = _dp(self.spectrum), _dp(self.lf0), _dp(self.lpf)
        return u


class Batch:
    """A batch of utterances resident in HBM (jb_batch_*).  `utts` are state-level utterances
    (Utterance), or -- with `pdf_set` -- pdf row indices (IndexUtterance: the per-state Gaussians are
    gathered and blended on the device, jb_batch_create_indexed), or parameter tracks (TrackUtterance:
    jb_batch_create_from_tracks, the run starts at the vocoder's frame prologue).  mlpg_only=True: the run
    ends with the parameter tracks (JB_BATCH_MLPG_ONLY; read them with track())."""

    def __init__(self, voice: VoiceInfo, utts: Sequence[Utterance], device: int = -1,
                 keep_tracks: bool = False, generic_mlpg: bool = False, serial: bool = False,
                 chunk_frames: int = 0, warmup_frames: int = 0, verify_tol: float = 0.0,
                 kernel: str = "auto", serial_gv: bool = False, pcm_i16: bool = False,
                 pdf_set: Optional[PdfSet] = None, mlpg_only: bool = False,
                 test_gang_timeout: bool = False, no_exc_table: bool = False):
        L = F.lib()
        self._L = L
        self.voice = voice
        self._utts = list(utts)  # keep host arrays alive during create
        vd, keep = voice.c_struct()
        from_tracks = bool(self._utts) and isinstance(self._utts[0], TrackUtterance)
        ty = F.TrackUtt if from_tracks else F.IndexUtt if pdf_set is not None else F.StateUtt
        arr = (ty * max(1, len(utts)))()
        for i, u in enumerate(self._utts):
            arr[i] = u.c_struct()
        opts = F.BatchOpts()
        opts.device = device
        opts.flags = ((F.BATCH_KEEP_TRACKS if keep_tracks else 0) | (F.BATCH_GENERIC_MLPG if generic_mlpg else 0)
                      | (F.BATCH_SERIAL if serial else 0) | (F.BATCH_SERIAL_GV if serial_gv else 0)
                      | (F.BATCH_PCM_I16 if pcm_i16 else 0) | (F.BATCH_MLPG_ONLY if mlpg_only else 0)
                      | (F.BATCH_TEST_GANG_TIMEOUT if test_gang_timeout else 0)
                      | (F.BATCH_NO_EXC_TABLE if no_exc_table else 0)
                      | {"auto": 0, "wave": F.BATCH_WAVE_KERNEL, "triple": F.BATCH_LANE_KERNEL}[kernel])
        opts.chunk_frames, opts.warmup_frames, opts.verify_tol = chunk_frames, warmup_frames, verify_tol
        self.flags, self.device = opts.flags | (F.BATCH_KEEP_TRACKS if mlpg_only else 0), device
        h = C.c_void_p()
        if from_tracks:
            F.check(L.jb_batch_create_from_tracks(C.byref(vd), arr, len(utts), C.byref(opts), C.byref(h)))
        elif pdf_set is not None:
            F.check(L.jb_batch_create_indexed(C.byref(vd), pdf_set._h, arr, len(utts), C.byref(opts), C.byref(h)))
        else:
            F.check(L.jb_batch_create(C.byref(vd), arr, len(utts), C.byref(opts), C.byref(h)))
        self._h = h
        self._keep = keep

    def __len__(self):
        return self._L.jb_batch_size(self._h)

    def run(self):
        F.check(self._L.jb_batch_run(self._h))

    def sync(self):
        F.check(self._L.jb_batch_sync(self._h))

    def run_timed(self):
        t, v = C.c_float(), C.c_float()
        F.check(self._L.jb_batch_run_timed(self._h, C.byref(t), C.byref(v)))
        return t.value, v.value

    def last_timing(self):
        """(total_ms, vocoder_kernel_ms) of the last completed run (after sync())."""
        t, v = C.c_float(), C.c_float()
        F.check(self._L.jb_batch_last_timing(self._h, C.byref(t), C.byref(v)))
        return t.value, v.value

    def info(self):
        """(chunk_frames, warmup_frames, n_items, n_redo) of the last run."""
        v = [C.c_uint32() for _ in range(4)]
        F.check(self._L.jb_batch_info(self._h, *[C.byref(x) for x in v]))
        return dict(chunk_frames=v[0].value, warmup_frames=v[1].value, n_items=v[2].value, n_redo=v[3].value)

    def kernel_info(self):
        """(kernel name, waves per SIMD) of the vocoder kernel the work list was built for ("k_vocoder" stands for
        its two-wave form k_vocoder_pair as well, which launches of at most two chunks per CU take)."""
        lt, w = C.c_uint32(), C.c_uint32()
        F.check(self._L.jb_batch_kernel_info(self._h, C.byref(lt), C.byref(w)))
        return ("k_vocoder_lt" if lt.value else "k_vocoder"), w.value

    def num_samples(self, i):
        return self._L.jb_batch_num_samples(self._h, i)

    def num_frames(self, i):
        return self._L.jb_batch_num_frames(self._h, i)

    @property
    def total_samples(self):
        return self._L.jb_batch_total_samples(self._h)

    def pcm(self, i) -> np.ndarray:
        n = self.num_samples(i)
        out = np.empty(n, dtype=np.float64)
        F.check(self._L.jb_batch_read_pcm(self._h, i, out.ctypes.data, n))
        return out

    def pcm_i16(self, i) -> np.ndarray:
        """16-bit PCM of a batch created with pcm_i16=True (fused clamp + truncate sink)."""
        n = self.num_samples(i)
        out = np.empty(n, dtype=np.int16)
        F.check(self._L.jb_batch_read_pcm_i16(self._h, i, out.ctypes.data, n))
        return out

    def pcm_all(self, out: Optional[List[np.ndarray]] = None) -> List[np.ndarray]:
        """Every utterance's PCM through the staged whole-slab read (jb_batch_read_pcm_all /
        _i16_all); `out` re-uses caller buffers (already touched pages: the read then runs at link rate)."""
        i16 = bool(self.flags & F.BATCH_PCM_I16)
        dt = np.int16 if i16 else np.float64
        n = len(self)
        if out is None:
            out = [np.empty(self.num_samples(i), dtype=dt) for i in range(n)]
        ptrs = (C.c_void_p * max(1, n))(*[o.ctypes.data if o.size else None for o in out])
        fn = self._L.jb_batch_read_pcm_i16_all if i16 else self._L.jb_batch_read_pcm_all
        F.check(fn(self._h, ptrs))
        return out

    def track(self, i, stream) -> np.ndarray:
        T, Lv = self.num_frames(i), self.voice.streams[stream].vector_length
        out = np.empty((T, Lv), dtype=np.float64)
        F.check(self._L.jb_batch_read_track(self._h, i, stream, out.ctypes.data, out.size))
        return out

    def coefficients(self, i) -> np.ndarray:
        """MLSA filter coefficients per frame, mc2b(postfilter_mcp(spectrum)) (vocoder/mod.rs:116-118)."""
        T, Lv = self.num_frames(i), self.voice.streams[0].vector_length
        out = np.empty((T, Lv), dtype=np.float64)
        F.check(self._L.jb_batch_read_coefficients(self._h, i, out.ctypes.data, out.size))
        return out

    def first_coefficients(self, i) -> np.ndarray:
        """The coefficients frame 0 starts from (un-filtered spectrum when a post-filter / stage is on)."""
        Lv = self.voice.streams[0].vector_length
        out = np.zeros(Lv, dtype=np.float64)
        F.check(self._L.jb_batch_read_first_coefficients(self._h, i, out.ctypes.data, out.size))
        return out

    def excitation(self, i) -> np.ndarray:
        n = self.num_samples(i)
        out = np.empty(n, dtype=np.float64)
        F.check(self._L.jb_batch_read_excitation(self._h, i, out.ctypes.data, n))
        return out

    def redo_stats(self):
        """(settled at the checkpoint, recomputed to the end) of the chunks that failed the hand-off check."""
        a, b = C.c_uint32(), C.c_uint32()
        F.check(self._L.jb_batch_redo_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def gang_fallbacks(self) -> int:
        """Times the resident GV kernel timed out in formation and the multi-launch sweeps took over."""
        return self._L.jb_batch_gang_fallbacks(self._h)

    def device_pcm(self):
        n = C.c_size_t()
        p = self._L.jb_batch_device_pcm(self._h, C.byref(n))
        return p, n.value

    def pcm_offset(self, i):
        return self._L.jb_batch_pcm_offset(self._h, i)

    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def mlpg_batch(voice: VoiceInfo, utts: Sequence[Utterance], device: int = -1, serial_gv: bool = False):
    """jb_mlpg_batch = MlpgAdjust::create for every stream of every utterance (mlpg_adjust/mod.rs:31-51):
    returns [[track of stream 0, 1, 2] per utterance], each [T][L] (NODATA in unvoiced LF0 frames)."""
    L = F.lib()
    vd, keep = voice.c_struct()
    arr = (F.StateUtt * max(1, len(utts)))()
    for i, u in enumerate(utts):
        arr[i] = u.c_struct()
    ns = len(voice.streams)
    nfr = (C.c_size_t * max(1, len(utts)))()
    opts = F.BatchOpts()
    opts.device, opts.flags = device, (F.BATCH_SERIAL_GV if serial_gv else 0)
    F.check(L.jb_mlpg_batch(C.byref(vd), arr, len(utts), C.byref(opts), None, nfr))
    out = [[np.empty((nfr[i], voice.streams[s].vector_length)) for s in range(ns)] for i in range(len(utts))]
    ptrs = (C.POINTER(C.c_double) * max(1, len(utts) * ns))(*[_dp(t) for u in out for t in u])
    F.check(L.jb_mlpg_batch(C.byref(vd), arr, len(utts), C.byref(opts), ptrs, nfr))
    del keep
    return out


def vocode_tracks_batch(voice: VoiceInfo, utts: Sequence["TrackUtterance"], device: int = -1):
    """jb_vocode_tracks_batch = SpeechGenerator::new + generate_all on given parameter tracks
    (speech.rs:25-50,87-96): a list of f64 PCM arrays."""
    L = F.lib()
    vd, keep = voice.c_struct()
    arr = (F.TrackUtt * max(1, len(utts)))()
    for i, u in enumerate(utts):
        arr[i] = u.c_struct()
    ns = (C.c_size_t * max(1, len(utts)))()
    opts = F.BatchOpts()
    opts.device = device
    F.check(L.jb_vocode_tracks_batch(C.byref(vd), arr, len(utts), C.byref(opts), None, ns))
    out = [np.empty(ns[i], dtype=np.float64) for i in range(len(utts))]
    ptrs = (C.POINTER(C.c_double) * max(1, len(utts)))(*[_dp(o) for o in out])
    F.check(L.jb_vocode_tracks_batch(C.byref(vd), arr, len(utts), C.byref(opts), ptrs, ns))
    del keep
    return out


def vocoder_synthesize_batch(voice: VoiceInfo, utts: Sequence["TrackUtterance"], device: int = -1):
    """jb_vocoder_synthesize_batch = Vocoder::new + Vocoder::synthesize per frame (vocoder/mod.rs:45-178): like
    vocode_tracks_batch without SpeechGenerator::new's LPF checks, so that nlpf == 0 (voice.streams[2]
    .vector_length == 0, lpf tracks of width 0) reaches Excitation::get's ring-buffer-less branch."""
    L = F.lib()
    vd, keep = voice.c_struct()
    arr = (F.TrackUtt * max(1, len(utts)))()
    for i, u in enumerate(utts):
        arr[i] = u.c_struct()
    ns = (C.c_size_t * max(1, len(utts)))()
    opts = F.BatchOpts()
    opts.device = device
    F.check(L.jb_vocoder_synthesize_batch(C.byref(vd), arr, len(utts), C.byref(opts), None, ns))
    out = [np.empty(ns[i], dtype=np.float64) for i in range(len(utts))]
    ptrs = (C.POINTER(C.c_double) * max(1, len(utts)))(*[_dp(o) for o in out])
    F.check(L.jb_vocoder_synthesize_batch(C.byref(vd), arr, len(utts), C.byref(opts), ptrs, ns))
    del keep
    return out


def generator_from_tracks(voice: VoiceInfo, utt: "TrackUtterance", device: int = -1):
    """jb_generator_new_from_tracks = SpeechGenerator::new on caller-held tracks (speech.rs:25-50); step it with
    generate_step / generate_steps / generate_all."""
    from .engine import SpeechGenerator

    L = F.lib()
    vd, keep = voice.c_struct()
    arr = (F.TrackUtt * 1)()
    arr[0] = utt.c_struct()
    opts = F.BatchOpts()
    opts.device = device
    h = C.c_void_p()
    F.check(L.jb_generator_new_from_tracks(C.byref(vd), arr, C.byref(opts), C.byref(h)))
    g = SpeechGenerator(h, L)
    g._keep = (keep, utt)
    return g


def paramgen_vocode_batch(voice: VoiceInfo, utts: Sequence[Utterance], device: int = -1,
                          devices: Optional[Sequence[int]] = None):
    """One-shot jb_paramgen_vocode_batch: returns a list of f64 PCM arrays.  With `devices` the batch
    goes through jb_paramgen_vocode_batch_multi: LPT split by frames over the listed GPUs, one host
    thread per device (a device may be listed more than once)."""
    if devices is None:
        with Batch(voice, utts, device=device) as b:
            b.run()
            b.sync()
            return [b.pcm(i) for i in range(len(utts))]
    L = F.lib()
    vd, keep = voice.c_struct()
    arr = (F.StateUtt * max(1, len(utts)))()
    for i, u in enumerate(utts):
        arr[i] = u.c_struct()
    fp = voice.fperiod
    out = [np.empty(int(u.durations.sum()) * fp, dtype=np.float64) for u in utts]
    ptrs = (C.POINTER(C.c_double) * max(1, len(utts)))(*[_dp(o) for o in out])
    ns = (C.c_size_t * max(1, len(utts)))()
    dv = (C.c_int32 * max(1, len(devices)))(*[int(d) for d in devices])
    F.check(L.jb_paramgen_vocode_batch_multi(C.byref(vd), arr, len(utts), None, dv, len(devices), ptrs, ns))
    assert all(ns[i] == len(out[i]) for i in range(len(utts)))
    del keep
    return out
