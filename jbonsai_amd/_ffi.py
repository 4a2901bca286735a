"""ctypes binding of libjbonsai_amd.so (include/jbonsai_amd.h).

The shared library is the product; this file only marshals arguments.  It
fails loudly when the HIP library is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libjbonsai_amd.so"

MAX_STREAM = 3
MAX_WINDOW = 8
NODATA = -1e10

JB_OK = 0
ERRORS = {
    -1: "JB_ERR_INVALID", -2: "JB_ERR_UNSUPPORTED", -3: "JB_ERR_DEVICE", -4: "JB_ERR_MODEL",
    -5: "JB_ERR_LABEL", -6: "JB_ERR_PARSE_OPTION", -7: "JB_ERR_WEIGHT", -8: "JB_ERR_BUFFER",
}
BATCH_KEEP_TRACKS = 1
BATCH_GENERIC_MLPG = 2
BATCH_SERIAL = 4
BATCH_WAVE_KERNEL = 8
BATCH_LANE_KERNEL = 16
BATCH_SERIAL_GV = 32
BATCH_PCM_I16 = 64
BATCH_MLPG_ONLY = 128
BATCH_TEST_GANG_TIMEOUT = 256
BATCH_NO_EXC_TABLE = 512


class JbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class StreamDesc(C.Structure):
    _fields_ = [
        ("vector_length", C.c_uint32),
        ("num_windows", C.c_uint32),
        ("is_msd", C.c_uint32),
        ("use_gv", C.c_uint32),
        ("win_width", C.c_uint32 * MAX_WINDOW),
        ("win_coef", C.POINTER(C.c_double)),
    ]


class VoiceDesc(C.Structure):
    _fields_ = [
        ("sampling_frequency", C.c_uint32),
        ("fperiod", C.c_uint32),
        ("nstream", C.c_uint32),
        ("stage", C.c_uint32),
        ("use_log_gain", C.c_uint32),
        ("alpha", C.c_double),
        ("beta", C.c_double),
        ("volume", C.c_double),
        ("stream", StreamDesc * MAX_STREAM),
    ]


class StreamStates(C.Structure):
    _fields_ = [
        ("mean", C.POINTER(C.c_double)),
        ("var", C.POINTER(C.c_double)),
        ("msd", C.POINTER(C.c_double)),
        ("gv_mean", C.POINTER(C.c_double)),
        ("gv_var", C.POINTER(C.c_double)),
        ("gv_switch", C.POINTER(C.c_uint8)),
        ("gv_weight", C.c_double),
        ("msd_threshold", C.c_double),
    ]


class StateUtt(C.Structure):
    _fields_ = [
        ("num_states", C.c_uint32),
        ("durations", C.POINTER(C.c_uint32)),
        ("stream", StreamStates * MAX_STREAM),
    ]


MAX_VOICES = 8


class PdfTable(C.Structure):
    _fields_ = [("rows", C.POINTER(C.c_float)), ("n_rows", C.c_uint32), ("row_len", C.c_uint32)]


class IndexStream(C.Structure):
    _fields_ = [
        ("row", C.POINTER(C.c_uint32) * MAX_VOICES),
        ("weight", C.POINTER(C.c_double)),
        ("gv_mean", C.POINTER(C.c_double)),
        ("gv_var", C.POINTER(C.c_double)),
        ("gv_switch", C.POINTER(C.c_uint8)),
        ("gv_weight", C.c_double),
        ("msd_threshold", C.c_double),
    ]


class IndexUtt(C.Structure):
    _fields_ = [
        ("num_states", C.c_uint32),
        ("durations", C.POINTER(C.c_uint32)),
        ("stream", IndexStream * MAX_STREAM),
        ("lf0_offset", C.c_double),
    ]


class TrackUtt(C.Structure):
    _fields_ = [
        ("n_spectrum", C.c_size_t), ("n_lf0", C.c_size_t), ("n_lpf", C.c_size_t),
        ("spectrum_width", C.c_uint32), ("lf0_width", C.c_uint32), ("lpf_width", C.c_uint32),
        ("reserved", C.c_uint32),
        ("spectrum", C.POINTER(C.c_double)), ("lf0", C.POINTER(C.c_double)), ("lpf", C.POINTER(C.c_double)),
    ]


class BatchOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("chunk_frames", C.c_uint32),
                ("warmup_frames", C.c_uint32), ("verify_tol", C.c_double), ("reserved0", C.c_uint32), ("reserved", C.c_uint32)]


# every symbol include/jbonsai_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "jb_batch_create", "jb_batch_run", "jb_batch_sync", "jb_batch_run_timed", "jb_batch_last_timing", "jb_batch_size",
    "jb_batch_num_frames", "jb_batch_num_samples", "jb_batch_total_samples", "jb_batch_read_pcm",
    "jb_batch_read_pcm_i16", "jb_batch_read_pcm_all", "jb_batch_read_pcm_i16_all", "jb_pdf_set_create", "jb_pdf_set_free", "jb_batch_create_indexed", "jb_batch_read_track", "jb_release_cached_memory", "jb_set_cached_memory_limit", "jb_batch_read_coefficients", "jb_batch_read_first_coefficients", "jb_batch_read_excitation", "jb_batch_device_pcm", "jb_batch_pcm_offset",
    "jb_batch_info", "jb_batch_kernel_info", "jb_batch_redo_stats", "jb_batch_gang_fallbacks", "jb_batch_free", "jb_paramgen_vocode_batch",
    "jb_mlpg_batch", "jb_batch_create_from_tracks", "jb_vocode_tracks_batch",
    "jb_engine_load", "jb_engine_load_from_bytes", "jb_engine_new", "jb_engine_free",
    "jb_engine_set_sampling_frequency", "jb_engine_get_sampling_frequency",
    "jb_engine_set_fperiod", "jb_engine_get_fperiod", "jb_engine_set_volume", "jb_engine_get_volume",
    "jb_engine_set_msd_threshold", "jb_engine_get_msd_threshold", "jb_engine_set_gv_weight",
    "jb_engine_get_gv_weight", "jb_engine_set_phoneme_alignment_flag",
    "jb_engine_get_phoneme_alignment_flag", "jb_engine_set_batch_invariant", "jb_engine_get_batch_invariant",
    "jb_engine_set_speed", "jb_engine_get_speed",
    "jb_engine_set_alpha", "jb_engine_get_alpha", "jb_engine_set_beta", "jb_engine_get_beta",
    "jb_engine_set_additional_half_tone", "jb_engine_get_additional_half_tone",
    "jb_engine_num_voices", "jb_engine_num_streams", "jb_engine_num_states",
    "jb_engine_set_interpolation_weight", "jb_engine_get_interpolation_weight", "jb_synthesize", "jb_pcm_free", "jb_write_wav_i16", "jb_write_wav_f64", "jb_synthesize_batch", "jb_synthesize_batch_i16", "jb_pcm_i16_free",
    "jb_engine_model_shape", "jb_engine_pdf_table", "jb_engine_tree_index",
    "jb_engine_states", "jb_states_utt", "jb_engine_voice_desc", "jb_states_free",
    "jb_generator_new", "jb_generator_new_from_tracks", "jb_vocoder_synthesize_batch", "jb_generator_fperiod", "jb_generator_synthesized_frames",
    "jb_generator_total_frames", "jb_generator_step", "jb_generator_step_n", "jb_generator_free",
    "jb_comm_unique_id", "jb_comm_init", "jb_comm_rank", "jb_comm_size", "jb_comm_free", "jb_gather_pcm",
    "jb_gathered_samples", "jb_gathered_sample_bytes", "jb_gathered_device", "jb_gathered_read", "jb_gathered_free",
    "jb_lpt_partition", "jb_paramgen_vocode_batch_multi", "jb_synthesize_batch_multi", "jb_synthesize_batch_i16_multi",
    "jb_states_duration_params", "jb_last_error", "jb_device_count", "jb_device_arch", "jb_device_pci_bus_id", "jb_version", "jb_default_verify_tol",
]


def build(force: bool = False) -> Path:
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = list((_HERE / "csrc").glob("*.hip")) + list((_HERE / "csrc").glob("*.cpp")) + \
        list((_HERE / "csrc").glob("*.h")) + [_HERE.parent / "include" / "jbonsai_amd.h"]
    stale = (not LIB_PATH.exists()) or any(p.stat().st_mtime > LIB_PATH.stat().st_mtime for p in srcs)
    if force or stale:
        r = subprocess.run(["bash", str(_HERE / "csrc" / "build.sh")], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc build failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def lib():
    """Load libjbonsai_amd.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                           " (the HIP library is the product; there is no fallback)")
    L = C.CDLL(str(LIB_PATH))
    vp, sz, dp = C.c_void_p, C.c_size_t, C.POINTER(C.c_double)
    L.jb_last_error.restype = C.c_char_p
    L.jb_version.restype = C.c_char_p
    L.jb_default_verify_tol.restype = C.c_double
    L.jb_device_count.restype = C.c_int
    L.jb_device_arch.argtypes = [C.c_int, C.c_char_p, sz]
    L.jb_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, sz]
    L.jb_batch_create.argtypes = [C.POINTER(VoiceDesc), C.POINTER(StateUtt), sz, C.POINTER(BatchOpts),
                                  C.POINTER(vp)]
    L.jb_pdf_set_create.argtypes = [C.POINTER(PdfTable), C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(vp)]
    L.jb_pdf_set_free.argtypes = [vp]
    L.jb_pdf_set_free.restype = None
    L.jb_batch_create_indexed.argtypes = [C.POINTER(VoiceDesc), vp, C.POINTER(IndexUtt), sz, C.POINTER(BatchOpts),
                                          C.POINTER(vp)]
    L.jb_batch_run.argtypes = [vp]
    L.jb_batch_sync.argtypes = [vp]
    L.jb_batch_run_timed.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.jb_batch_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for n in ("jb_batch_size", "jb_batch_total_samples"):
        getattr(L, n).restype = sz
        getattr(L, n).argtypes = [vp]
    for n in ("jb_batch_num_frames", "jb_batch_num_samples", "jb_batch_pcm_offset"):
        getattr(L, n).restype = sz
        getattr(L, n).argtypes = [vp, sz]
    L.jb_batch_read_pcm.argtypes = [vp, sz, vp, sz]
    L.jb_batch_read_pcm_all.argtypes = [vp, C.POINTER(vp)]
    L.jb_batch_read_pcm_i16_all.argtypes = [vp, C.POINTER(vp)]
    L.jb_batch_read_track.argtypes = [vp, sz, C.c_uint32, vp, sz]
    L.jb_batch_read_coefficients.argtypes = [vp, sz, vp, sz]
    L.jb_batch_read_first_coefficients.argtypes = [vp, sz, vp, sz]
    L.jb_batch_read_pcm_i16.argtypes = [vp, sz, vp, sz]
    L.jb_batch_read_excitation.argtypes = [vp, sz, vp, sz]
    L.jb_batch_device_pcm.restype = vp
    L.jb_batch_device_pcm.argtypes = [vp, C.POINTER(sz)]
    L.jb_batch_info.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 4
    L.jb_batch_redo_stats.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.jb_batch_kernel_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.jb_batch_gang_fallbacks.argtypes = [vp]
    L.jb_batch_gang_fallbacks.restype = C.c_uint32
    L.jb_batch_free.argtypes = [vp]
    L.jb_batch_free.restype = None
    L.jb_paramgen_vocode_batch.argtypes = [C.POINTER(VoiceDesc), C.POINTER(StateUtt), sz,
                                           C.POINTER(BatchOpts), C.POINTER(dp), C.POINTER(sz)]
    L.jb_mlpg_batch.argtypes = [C.POINTER(VoiceDesc), C.POINTER(StateUtt), sz, C.POINTER(BatchOpts),
                                C.POINTER(dp), C.POINTER(sz)]
    L.jb_batch_create_from_tracks.argtypes = [C.POINTER(VoiceDesc), C.POINTER(TrackUtt), sz, C.POINTER(BatchOpts),
                                              C.POINTER(vp)]
    L.jb_vocode_tracks_batch.argtypes = [C.POINTER(VoiceDesc), C.POINTER(TrackUtt), sz, C.POINTER(BatchOpts),
                                         C.POINTER(dp), C.POINTER(sz)]
    L.jb_vocoder_synthesize_batch.argtypes = [C.POINTER(VoiceDesc), C.POINTER(TrackUtt), sz, C.POINTER(BatchOpts),
                                              C.POINTER(dp), C.POINTER(sz)]
    L.jb_generator_new_from_tracks.argtypes = [C.POINTER(VoiceDesc), C.POINTER(TrackUtt), C.POINTER(BatchOpts),
                                               C.POINTER(vp)]
    L.jb_set_cached_memory_limit.argtypes = [sz]
    L.jb_comm_unique_id.argtypes = [C.c_char_p, sz]
    L.jb_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int32, C.POINTER(vp)]
    L.jb_comm_rank.argtypes = [vp]
    L.jb_comm_size.argtypes = [vp]
    L.jb_comm_free.argtypes = [vp]
    L.jb_comm_free.restype = None
    L.jb_gather_pcm.argtypes = [vp, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_float)]
    L.jb_gathered_samples.argtypes = [vp, C.c_int]
    L.jb_gathered_samples.restype = sz
    L.jb_gathered_sample_bytes.argtypes = [vp]
    L.jb_gathered_sample_bytes.restype = sz
    L.jb_gathered_device.argtypes = [vp, C.c_int]
    L.jb_gathered_device.restype = vp
    L.jb_gathered_read.argtypes = [vp, C.c_int, vp, sz]
    L.jb_gathered_free.argtypes = [vp]
    L.jb_gathered_free.restype = None
    L.jb_lpt_partition.argtypes = [C.POINTER(C.c_uint64), sz, sz, C.POINTER(C.c_uint32)]
    L.jb_paramgen_vocode_batch_multi.argtypes = [C.POINTER(VoiceDesc), C.POINTER(StateUtt), sz, C.POINTER(BatchOpts),
                                                 C.POINTER(C.c_int32), sz, C.POINTER(dp), C.POINTER(sz)]
    L.jb_write_wav_i16.argtypes = [C.c_char_p, vp, sz, C.c_uint32]
    L.jb_write_wav_f64.argtypes = [C.c_char_p, vp, sz, C.c_uint32]
    _lib = L
    return L


def write_wav(path, pcm, sampling_frequency: int) -> None:
    """16-bit mono WAV as the reference's examples write it (examples/is-bonsai/main.rs:37-49).
    int16 samples are written as they are; float64 samples are clamped and truncated first."""
    import numpy as np

    a = np.ascontiguousarray(pcm)
    if a.dtype == np.int16:
        check(lib().jb_write_wav_i16(str(path).encode(), a.ctypes.data, a.size, sampling_frequency))
    else:
        a = np.ascontiguousarray(a, dtype=np.float64)
        check(lib().jb_write_wav_f64(str(path).encode(), a.ctypes.data, a.size, sampling_frequency))


def check(rc):
    if rc != JB_OK:
        raise JbError(rc, (lib().jb_last_error() or b"").decode(errors="replace"))
