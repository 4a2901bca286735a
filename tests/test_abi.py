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
    from tests.helpers import PCM_TOL, VERIFY_TOL

    # the ONE tolerance of the PCM gates (tests/helpers.py) is the library's own hand-off tolerance
    assert L.jb_default_verify_tol() == VERIFY_TOL and PCM_TOL == 2 * VERIFY_TOL
    assert L.jb_device_count() >= 0


def test_struct_layouts_header_vs_ctypes(tmp_path):
    """Every struct that crosses the boundary: the header's compile-time layout table (sizes and key offsets,
    `JB_LAYOUT_ASSERT` at the end of include/jbonsai_amd.h, compiled here as C11 and as C++17) against the
    ctypes mirror the tests call through.  A Rust `#[repr(C)]` mirror has the same layout."""
    import ctypes as C
    import subprocess

    src = tmp_path / "lay.c"
    names = ["jb_stream_desc", "jb_voice_desc", "jb_stream_states", "jb_state_utt", "jb_batch_opts", "jb_pdf_table",
             "jb_index_stream", "jb_index_utt", "jb_track_utt"]
    src.write_text('#include "jbonsai_amd.h"\n#include <stdio.h>\nint main(void){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    sizes = None
    for cc, std, lang in (("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "c++")):
        exe = tmp_path / ("lay_" + cc.replace("+", "p"))
        subprocess.run([cc, std, "-x", lang, "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
        got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
        assert sizes is None or sizes == got
        sizes = got
    mirror = [_ffi.StreamDesc, _ffi.VoiceDesc, _ffi.StreamStates, _ffi.StateUtt, _ffi.BatchOpts, _ffi.PdfTable,
              _ffi.IndexStream, _ffi.IndexUtt, _ffi.TrackUtt]
    assert sizes == [C.sizeof(m) for m in mirror] == [56, 216, 64, 208, 32, 16, 112, 360, 64]
    assert _ffi.VoiceDesc.alpha.offset == 24 and _ffi.VoiceDesc.stream.offset == 48
    assert _ffi.StateUtt.stream.offset == 16 and _ffi.IndexUtt.lf0_offset.offset == 352
    assert _ffi.TrackUtt.spectrum.offset == 40 and _ffi.BatchOpts.verify_tol.offset == 16


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


def test_library_asks_for_hardware_queues_in_a_fresh_process():
    """Loading the library sets GPU_MAX_HW_QUEUES (16) unless the host has set it: streams that share one of the
    HIP runtime's default four hardware queues run one after the other, and batches in flight need their own
    (INTEGRATION.md section 5).  Checked in child processes: the C environment of this one is long set."""
    import os
    import subprocess
    import sys

    code = ("import ctypes, sys; ctypes.CDLL(sys.argv[1]); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "print(libc.getenv(b'GPU_MAX_HW_QUEUES'))")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", code, str(J.LIB_PATH)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "b'16'" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code, str(J.LIB_PATH)], capture_output=True, text=True,
                       env=dict(env, GPU_MAX_HW_QUEUES="6"), timeout=120)
    assert r.returncode == 0 and "b'6'" in r.stdout, r.stdout + r.stderr
    # opt-out: JB_LEAVE_HIP_ENV=1 -- the library leaves the process's environment alone
    r = subprocess.run([sys.executable, "-c", code, str(J.LIB_PATH)], capture_output=True, text=True,
                       env=dict(env, JB_LEAVE_HIP_ENV="1"), timeout=120)
    assert r.returncode == 0 and "None" in r.stdout, r.stdout + r.stderr


def test_index_utterance_struct_cache_follows_assignments():
    """IndexUtterance.c_struct() is marshalled once per object: assigning a field of the utterance or of
    one of its streams must drop the cached struct, or the next Batch silently uploads the old arrays."""
    import numpy as np

    from jbonsai_amd.batch import IndexStreamStates, IndexUtterance

    st = IndexStreamStates(rows=[np.arange(5, dtype=np.uint32)], weights=np.ones(1))
    u = IndexUtterance(durations=np.full(5, 3, dtype=np.uint32), streams=[st], lf0_offset=0.0)
    c0 = u.c_struct()
    assert u.c_struct() is c0 and c0.num_states == 5
    u.durations = np.full(7, 2, dtype=np.uint32)
    c1 = u.c_struct()
    assert c1 is not c0 and c1.num_states == 7 and c1.durations[0] == 2
    new_rows = np.arange(10, 17, dtype=np.uint32)
    st.rows = [new_rows]
    c2 = u.c_struct()
    assert c2 is not c1 and c2.stream[0].row[0][0] == 10
    assert u.c_struct() is c2
