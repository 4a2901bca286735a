"""Post-build ISA check of the dominant kernel (no GPU needed: hipcc cross-compiles).

k_vocoder_lt issues its excitation loads from inline asm and waits for them with hand-written counted
`s_waitcnt vmcnt`; the compiler does not know those VGPRs are in flight.  tools/asm_xload_check.py verifies on
the generated ISA that nothing reads or writes a load's destination between the load and the wait -- the gate
to pass at every toolchain change."""
import importlib.util
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = "/opt/rocm/bin/hipcc"


def _checker():
    spec = importlib.util.spec_from_file_location("asm_xload_check", ROOT / "tools" / "asm_xload_check.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_checker_sees_a_violation():
    m = _checker()
    ok = """
_Z3k_vocoder_ltv:
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_mul_f64 v[10:11], v[4:5], v[8:9]
	;;#ASMSTART
	global_load_dwordx2 v[4:5], v[20:21], off
	;;#ASMEND
	v_fma_f64 v[12:13], v[10:11], v[10:11], v[12:13]
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""
    r = m.check(ok)
    assert list(r.values()) == [(1, [])]
    bad = ok.replace("v_fma_f64 v[12:13], v[10:11], v[10:11], v[12:13]", "v_mov_b32_e32 v5, v30")
    (n, viol), = m.check(bad).values()
    assert n == 1 and len(viol) == 1 and "touches the destination" in viol[0]
    # a range operand that covers the destination counts as well
    bad2 = ok.replace("v_fma_f64 v[12:13], v[10:11], v[10:11], v[12:13]", "global_store_dwordx4 v[20:21], v[2:5], off")
    assert len(list(m.check(bad2).values())[0][1]) == 1


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_vocoder_lt_asm_loads_are_untouched_until_their_wait(tmp_path):
    src = ROOT / "jbonsai_amd" / "csrc" / "jb_vocoder.hip"
    out = tmp_path / "voc.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                    "--cuda-device-only", "-S", str(src), "-o", str(out)], check=True, capture_output=True,
                   cwd=src.parent)
    res = _checker().check(out.read_text())
    # the two exact forms (35, 25) + the three lane-triple codes (25, 31, 35) and the three one-stage-per-lane codes
    # (41, 51, 61) that run every other order, each at W = 8 and 4 waves per workgroup
    assert len(res) == 16, list(res)
    for k, (n, viol) in res.items():
        assert n == 3, (k, n)       # the first odd request, then one request per even / odd sample
        assert not viol, (k, viol)
