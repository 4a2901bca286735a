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


def test_no_torch_or_oracle_in_product():
    """The product must not import torch types into its ABI nor touch oracle/."""
    for p in (ROOT / "jbonsai_amd").rglob("*"):
        if p.suffix in {".py", ".cpp", ".hip", ".h"}:
            s = p.read_text()
            assert "oracle" not in s.replace("the oracle", "").replace("CPU oracle", "") or p.name == "jb_mlpg.hip", p
            assert "import torch" not in s, p
