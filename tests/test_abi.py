import pytest
"""The C-ABI library loads and exports every symbol include/jbonsai_amd.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import re
from pathlib import Path

import jbonsai_amd as J
from jbonsai_amd import _ffi

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "jbonsai_amd.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(jb_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(str(J.LIB_PATH))
    syms = declared_symbols()
    assert len(syms) >= 55
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_lists_every_symbol():
    assert sorted(_ffi.SYMBOLS) == declared_symbols()


def test_version_and_no_gpu_error_path():
    L = J.lib()
    assert b"gfx950" in L.jb_version()
    assert L.jb_device_count() >= 0


def test_no_link_time_dependency_on_rccl_or_torch():
    """RCCL is bound with dlopen at the first gather; the library links neither it nor torch."""
    import subprocess

    out = subprocess.run(["ldd", str(_ffi.LIB_PATH)], capture_output=True, text=True).stdout
    assert "rccl" not in out and "nccl" not in out and "torch" not in out and "libjbo_oracle" not in out, out


def test_no_torch_or_oracle_in_product():
    """The product must not import torch types into its ABI nor touch oracle/."""
    for p in (ROOT / "jbonsai_amd").rglob("*"):
        if p.suffix in {".py", ".cpp", ".hip", ".h"}:
            s = p.read_text()
            assert "oracle" not in s.replace("the oracle", "").replace("CPU oracle", "") or p.name == "jb_mlpg.hip", p
            assert "import torch" not in s, p


def test_wav_sink_roundtrip(tmp_path):
    """jb_write_wav_{i16,f64}: 16-bit mono RIFF as hound writes it in the reference's examples
    (examples/is-bonsai/main.rs:37-49); f64 goes through min/max clamp + `as i16` truncation."""
    import wave

    import numpy as np

    import jbonsai_amd as J

    x = np.array([0.0, 0.9, -0.9, 1.5, -1.5, 32767.4, 32768.0, 1e9, -32768.9, -1e9, 12345.678], dtype=np.float64)
    want = np.clip(x, -32768.0, 32767.0).astype(np.int16)
    for name, data in (("f.wav", x), ("i.wav", want)):
        p_ = tmp_path / name
        J.write_wav(p_, data, 48000)
        with wave.open(str(p_), "rb") as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 48000, len(x))
            got = np.frombuffer(w.readframes(len(x)), dtype="<i2")
        assert np.array_equal(got, want)
    with pytest.raises(J.JbError):
        J.write_wav(tmp_path / "no_such_dir" / "x.wav", want, 48000)
