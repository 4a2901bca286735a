"""jbonsai_amd: MI355X (gfx950) implementation of jbonsai's parameter-generation +
MLSA-vocoder hot path behind a C ABI (include/jbonsai_amd.h).

The HIP shared library `libjbonsai_amd.so` is the product; this package is the
thin host-side mirror of the reference's API used by tests and the benchmark.
"""
from ._ffi import JbError, LIB_PATH, NODATA, build, lib, write_wav  # noqa: F401
from .batch import (Batch, IndexStreamStates, IndexUtterance, PdfSet, StreamInfo, StreamStates, TrackUtterance,  # noqa: F401
                    Utterance, VoiceInfo, generator_from_tracks, mlpg_batch, paramgen_vocode_batch, vocode_tracks_batch,
                    vocoder_synthesize_batch)

from .engine import Engine, SpeechGenerator  # noqa: F401,E402
from . import comm  # noqa: F401,E402

__all__ = ["Engine", "SpeechGenerator", "JbError", "LIB_PATH", "NODATA", "build", "lib", "write_wav", "Batch", "StreamInfo", "StreamStates",
           "Utterance", "VoiceInfo", "paramgen_vocode_batch", "mlpg_batch", "vocode_tracks_batch", "vocoder_synthesize_batch", "generator_from_tracks", "TrackUtterance", "PdfSet", "IndexUtterance", "IndexStreamStates"]
