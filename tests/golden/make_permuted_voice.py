#!/usr/bin/env python3
"""Derives a SECOND voice from the nitech voice for the two-voice configurations
(BASELINE configs 4/5; SURVEY.md section 8d: "substitute second voice = nitech pdf
tables re-indexed by a seeded permutation of leaves, same metadata").

The reference's own two-voice golden (`bonsai_multi`, /root/reference/src/lib.rs:77-91)
needs the tohoku-f01 files, which are an un-fetched submodule of the reference tree.
What a two-voice test must be able to see is a voice-index or weight mix-up in the
gather/blend (VoiceSet::weighted, src/model/voice_set.rs:80-95), and for that the two
tables have to DIFFER.  This script rewrites only the pdf blobs of the input voice:

  * in every tree of every model (duration, MCP, LF0, LPF, GV-MCP, GV-LF0) the pdf rows are
    permuted (Fisher-Yates driven by splitmix64, seed below ^ model ordinal ^ tree; never the identity);
  * a tree with a single pdf (the five LPF trees) cannot be permuted: its means are scaled
    by 0.75 instead, so that the third stream differs between the voices as well.

Header, windows, questions and trees are byte-identical, so the result passes
VoiceSet::new's metadata checks (src/model/voice_set.rs:22-42) and every label reaches a
leaf of the same tree with the same 1-based index -- which now holds another row.

The output is deterministic; tests and bench.py generate it on the fly (nothing but the
CC-BY nitech file under tests/golden/voice/ is read), so it is not committed.

Usage: make_permuted_voice.py [in.htsvoice] [out.htsvoice]
"""
from __future__ import annotations

import re
import struct
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
NITECH = HERE / "voice" / "nitech_jp_atr503_m001.htsvoice"
SEED = 0x7065726D75746564  # "permuted"
MASK = (1 << 64) - 1
LPF_SCALE = 0.75


class _SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)


def _permutation(n, seed):
    rng = _SplitMix64(seed)
    p = list(range(n))
    for i in range(n - 1, 0, -1):
        j = rng.next() % (i + 1)
        p[i], p[j] = p[j], p[i]
    if n > 1 and p == list(range(n)):  # a table of two or three rows may draw the identity: rotate instead
        p = p[1:] + p[:1]
    return p


def _header(raw: bytes):
    cut = raw.index(b"[DATA]\n") + len(b"[DATA]\n")
    text = raw[:cut].decode("ascii")
    kv = {}
    for ln in text.splitlines():
        if ":" in ln and not ln.startswith("["):
            k, v = ln.split(":", 1)
            kv[k] = v
    return cut, kv


def permuted_voice_bytes(raw: bytes, seed: int = SEED) -> bytes:
    cut, kv = _header(raw)
    blob = bytearray(raw[cut:])
    nstate = int(kv["NUM_STATES"])
    streams = kv["STREAM_TYPE"].split(",")
    models = [("DURATION_PDF", 2 * nstate, 0)]
    for s in streams:
        L, W = int(kv[f"VECTOR_LENGTH[{s}]"]), int(kv[f"NUM_WINDOWS[{s}]"])
        msd = int(kv[f"IS_MSD[{s}]"])
        models.append((f"STREAM_PDF[{s}]", 2 * L * W + msd, L * W))
        if int(kv[f"USE_GV[{s}]"]):
            models.append((f"GV_PDF[{s}]", 2 * L, L))
    for ordinal, (key, pdf_len, n_mean) in enumerate(models):
        lo, hi = (int(x) for x in re.fullmatch(r"(\d+)-(\d+)", kv[key]).groups())
        size = hi - lo + 1  # POSITION ranges are inclusive (src/model/parser/mod.rs:176-187)
        # layout (src/model/parser/model/mod.rs:38-60): ntree x u32 npdf, then per tree npdf x pdf_len f32;
        # ntree itself is in the tree text: take the one count that accounts for every byte of the blob
        ntree = None
        for nt in range(1, 65):
            if 4 * nt > size:
                break
            cnt = struct.unpack_from(f"<{nt}I", blob, lo)
            if 4 * nt + 4 * pdf_len * sum(cnt) == size:
                ntree, npdf = nt, cnt
                break
        if ntree is None:
            raise ValueError(f"{key}: cannot account for {size} bytes with rows of {pdf_len} f32")
        off = lo + 4 * ntree
        row_b = 4 * pdf_len
        for t, n in enumerate(npdf):
            rows = [bytes(blob[off + r * row_b: off + (r + 1) * row_b]) for r in range(n)]
            if n > 1:
                perm = _permutation(n, seed ^ (ordinal << 32) ^ t)
                rows = [rows[perm[r]] for r in range(n)]
            elif n_mean:
                v = list(struct.unpack(f"<{pdf_len}f", rows[0]))
                for k in range(n_mean):
                    v[k] = struct.unpack("<f", struct.pack("<f", v[k] * LPF_SCALE))[0]
                rows = [struct.pack(f"<{pdf_len}f", *v)]
            blob[off: off + n * row_b] = b"".join(rows)
            off += n * row_b
        assert off == hi + 1
    return raw[:cut] + bytes(blob)


def permuted_voice_path(dst_dir, src: Path = NITECH) -> Path:
    """Writes (once) and returns <dst_dir>/nitech_permuted.htsvoice."""
    dst = Path(dst_dir) / "nitech_permuted.htsvoice"
    data = permuted_voice_bytes(Path(src).read_bytes())
    if not dst.exists() or dst.read_bytes() != data:
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_bytes(data)
    return dst


if __name__ == "__main__":
    src = Path(sys.argv[1]) if len(sys.argv) > 1 else NITECH
    out = Path(sys.argv[2]) if len(sys.argv) > 2 else Path("nitech_permuted.htsvoice")
    out.write_bytes(permuted_voice_bytes(src.read_bytes()))
    print(f"wrote {out} ({out.stat().st_size} bytes)")
