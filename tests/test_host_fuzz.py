"""The front half takes bytes from outside the process: voice files (Engine::load / load_from_bytes,
src/model/voice/parser) and full-context labels (Engine::synthesize).  Damaged input must come back as an error
(ModelError / LabelError in the reference) or as a voice that still answers -- never as a crash, an over-read or a
hang.  Seeded mutations of the nitech voice and of the reference's sample labels; CPU only.  `tools/asan_cpu.sh`
runs this file with the library's host code under AddressSanitizer + UBSan."""
import os
import re

import numpy as np
import pytest

import jbonsai_amd as J
from tests.conftest import VOICE
from tests.golden.labels import SAMPLE_SENTENCE_1


SCALE = int(os.environ.get("JB_FUZZ_SCALE", "1"))  # tools/asan_cpu.sh: JB_FUZZ_SCALE=10 for a longer sanitizer run


@pytest.fixture(scope="module")
def voice_bytes():
    return VOICE.read_bytes()


def _try_load_and_use(raw):
    """load -> metadata -> states of a sentence -> close; J.JbError anywhere is a pass."""
    try:
        e = J.Engine.load_from_bytes([raw])
    except J.JbError:
        return "refused"
    try:
        vi = e.voice_info()
        assert len(vi.streams) >= 1  # (FRAME_PERIOD:0 loads, here as in the reference; a batch refuses it)
        try:
            u = e.states(SAMPLE_SENTENCE_1)
            assert np.all(np.isfinite(u.durations))
        except J.JbError:
            return "loaded, labels refused"
        return "loaded"
    finally:
        e.close()


def test_truncated_voices(voice_bytes):
    """Every section boundary of the file and a spread of cuts inside the sections."""
    head = voice_bytes[:voice_bytes.find(b"[DATA]")].decode()
    data0 = voice_bytes.find(b"[DATA]") + len(b"[DATA]\n")
    cuts = {0, 1, 7, 100, data0 - 1, data0, data0 + 1, len(voice_bytes) - 1}
    for lo, hi in re.findall(r"(\d+)-(\d+)", head):
        for p in (int(lo), int(hi)):
            cuts.update({data0 + p - 1, data0 + p, data0 + p + 1})
    rng = np.random.default_rng(5)
    cuts.update(int(x) for x in rng.integers(0, len(voice_bytes), 24 * SCALE))
    outcomes = {}
    for c in sorted(x for x in cuts if 0 <= x < len(voice_bytes)):
        r = _try_load_and_use(voice_bytes[:c])
        outcomes[r] = outcomes.get(r, 0) + 1
    assert outcomes.get("refused", 0) > 0  # (a cut inside the header or before the last section cannot load)


def test_header_mutations(voice_bytes):
    """Numbers of the text header replaced by hostile ones (zero, huge, negative, reversed ranges, past the end),
    keys dropped or doubled, separators broken."""
    i = voice_bytes.find(b"[DATA]")
    head, rest = voice_bytes[:i].decode(), voice_bytes[i:]
    lines = head.split("\n")
    rng = np.random.default_rng(11)
    hostile = ["0", "-1", "4294967295", "4294967296", "18446744073709551615", "99999999999999999999", "1e9", "", "x",
               "2147483648", "65536"]
    variants = []
    for k, ln in enumerate(lines):
        if ":" not in ln:
            continue
        key, val = ln.split(":", 1)
        for h in hostile:
            variants.append(lines[:k] + [key + ":" + h] + lines[k + 1:])
        a = re.findall(r"(\d+)-(\d+)", val)
        if a:  # a range list: reversed, overlapping, past the end
            lo, hi = a[0]
            for newval in (f"{hi}-{lo}", f"{lo}-{len(voice_bytes) * 2}", f"{lo}-{lo}", f"{hi}-{hi}", f"{lo}-",
                           f"-{hi}", f"{lo}-{hi}," * 40 + f"{lo}-{hi}", f"{int(hi) + 1}-{int(hi) + 2}"):
                variants.append(lines[:k] + [key + ":" + newval] + lines[k + 1:])
        variants.append(lines[:k] + lines[k + 1:])          # key missing
        variants.append(lines[:k] + [ln, ln] + lines[k + 1:])  # key twice
        variants.append(lines[:k] + [key + val] + lines[k + 1:])  # separator gone
    # a seeded sample (the full product is ~1,500 loads of a 1.2 MB file)
    pick = rng.choice(len(variants), size=min(800 * SCALE, len(variants)), replace=False)
    n_refused = 0
    for p in pick:
        raw = "\n".join(variants[int(p)]).encode() + rest
        n_refused += _try_load_and_use(raw) == "refused"
    assert n_refused > 0


def test_data_mutations(voice_bytes):
    """Bytes of the binary sections flipped: tree text (questions, node numbers), pdf counts, window lengths."""
    i = voice_bytes.find(b"[DATA]") + len(b"[DATA]\n")
    head = voice_bytes[:i].decode()
    spans = [(int(lo), int(hi)) for lo, hi in re.findall(r"(\d+)-(\d+)", head)]
    rng = np.random.default_rng(17)
    for trial in range(400 * SCALE):
        raw = bytearray(voice_bytes)
        lo, hi = spans[int(rng.integers(0, len(spans)))]
        # the first bytes of a section (counts, the first question / node line) matter most; then anywhere in it
        for _ in range(int(rng.integers(1, 6))):
            off = lo + int(rng.integers(0, min(64, hi - lo + 1))) if rng.random() < 0.6 else int(rng.integers(lo, hi + 1))
            raw[i + off] = int(rng.integers(0, 256))
        _try_load_and_use(bytes(raw))


def test_label_mutations():
    """Full-context labels with fields cut, doubled, emptied, very long, with foreign bytes: LabelError or states."""
    e = J.Engine.load([VOICE])
    rng = np.random.default_rng(23)
    base = list(SAMPLE_SENTENCE_1)
    seps = list("^-+=/:_|@!#%&[]")
    for trial in range(1000 * SCALE):
        labs = list(base)
        k = int(rng.integers(0, len(labs)))
        s = labs[k]
        kind = int(rng.integers(0, 8))
        if kind == 0:
            s = s[:int(rng.integers(0, len(s)))]
        elif kind == 1:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + s[p:] * 2
        elif kind == 2:
            s = "".join(c for c in s if c != seps[int(rng.integers(0, len(seps)))])
        elif kind == 3:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + "9" * 5000 + s[p:]
        elif kind == 4:
            s = "123 456 " + s if rng.random() < 0.5 else "456 123 " + s
        elif kind == 5:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + "\tあ\x7f" + s[p:]
        elif kind == 6:
            s = s.replace("xx", "", 3)
        else:
            s = "".join(seps[int(x)] for x in rng.integers(0, len(seps), 200))
        labs[k] = s
        try:
            u = e.states(labs)
            assert np.all(np.isfinite(u.durations)) and u.durations.min() >= 0
        except J.JbError as err:
            assert err.code in (-5, -1)
    e.close()
