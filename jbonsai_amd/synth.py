"""Synthetic state-level workloads built from the loaded voice's real pdfs.

The reference's benchmark input (the ~128 s preamble label file) is not in the
reference tree, so BASELINE config 2/3 run on seeded
synthetic utterances: random phones whose per-state Gaussians are real leaves of
the voice's decision trees; durations come from real duration pdfs.
PRNG: splitmix64, seed 0x6A626F6E73616921 ^ utterance_id."""
from __future__ import annotations

from typing import List

import numpy as np

from .batch import IndexStreamStates, IndexUtterance, PdfSet, StreamStates, Utterance
from .engine import Engine

MASK = (1 << 64) - 1
SEED = 0x6A626F6E73616921
T_128S = 25546  # frames of the ~128 s utterance (6,131,040 samples at 48 kHz / 240)


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n


class VoiceTables:
    """pdf tables of voice 0, pulled through the C ABI (jb_engine_pdf_table)."""

    def __init__(self, eng: Engine):
        self.eng = eng
        self.vi = eng.voice_info()
        self.nstate = eng.num_states
        self.dur = eng.pdf_table(0, 0)
        self.stream = []
        self.gv = []
        for s in range(eng.num_streams):
            nt, _ = eng.model_shape(1 + s)
            self.stream.append([eng.pdf_table(1 + s, t) for t in range(nt)])
            if self.vi.streams[s].use_gv:
                self.gv.append(eng.pdf_table(4 + s, 0))
            else:
                self.gv.append(None)
        # all trees of a stream concatenated (the layout of jb_pdf_table) and each tree's first row
        self.concat = [np.concatenate(t, axis=0) for t in self.stream]
        self.tree_off = [np.cumsum([0] + [len(x) for x in t[:-1]]) for t in self.stream]

    def pdf_set(self, device: int = -1) -> PdfSet:
        """The voice's stream tables resident on `device` (for indexed batches)."""
        return PdfSet([self.concat], device=device)


def synth_utterance(tab: VoiceTables, target_frames: int, utt_id: int = 0, indexed: bool = False):
    """indexed=False: state-level Utterance (per-state Gaussians expanded on the host);
    indexed=True: the SAME utterance as pdf row indices (IndexUtterance) for a batch created over
    VoiceTables.pdf_set() -- 12 bytes per state cross PCIe instead of 2.2 kB."""
    rng = SplitMix64(SEED ^ utt_id)
    ns = tab.nstate
    nstream = len(tab.stream)
    durs: List[int] = []
    rows = [[] for _ in range(nstream)]
    ridx = [[] for _ in range(nstream)]
    total = 0
    while total < target_frames:
        dp = tab.dur[rng.below(len(tab.dur))]
        leaf = [[rng.below(len(tab.stream[s][min(k, len(tab.stream[s]) - 1)])) for k in range(ns)]
                for s in range(nstream)]
        for k in range(ns):
            d = max(1, int(np.floor(float(dp[k]) + 0.5)))
            if total + d > target_frames:
                d = target_frames - total
            if d <= 0:
                break
            durs.append(d)
            total += d
            for s in range(nstream):
                tk = min(k, len(tab.stream[s]) - 1)
                if indexed:
                    ridx[s].append(int(tab.tree_off[s][tk]) + leaf[s][k])
                else:
                    rows[s].append(tab.stream[s][tk][leaf[s][k]])
            if total >= target_frames:
                break
    S = len(durs)
    nphone = (S + ns - 1) // ns
    sw = np.ones(S, dtype=np.uint8)
    sw[:ns] = 0                      # first / last phone behave like silence: GV off
    sw[(nphone - 1) * ns:] = 0
    streams = []
    if indexed:
        one = np.ones(1)
        for s in range(nstream):
            si = tab.vi.streams[s]
            gm = gv = gs = None
            if si.use_gv:
                g = tab.gv[s][0].astype(np.float64)
                gm, gv, gs = g[:si.vector_length], g[si.vector_length:], sw
            streams.append(IndexStreamStates([np.asarray(ridx[s], dtype=np.uint32)], one, gm, gv, gs, 1.0, 0.5))
        return IndexUtterance(np.asarray(durs, dtype=np.uint32), streams)
    for s in range(nstream):
        si = tab.vi.streams[s]
        WL = si.vector_length * len(si.windows)
        arr = np.asarray(rows[s], dtype=np.float64).reshape(S, -1)
        mean, var = arr[:, :WL], arr[:, WL:2 * WL]
        msd = arr[:, 2 * WL] if si.is_msd else None
        gm = gv = gs = None
        if si.use_gv:
            g = tab.gv[s][0].astype(np.float64)
            gm, gv, gs = g[:si.vector_length], g[si.vector_length:], sw
        streams.append(StreamStates(mean, var, msd, gm, gv, gs, 1.0, 0.5))
    return Utterance(np.asarray(durs, dtype=np.uint32), streams)


def u128(tab: VoiceTables, utt_id: int = 0) -> Utterance:
    """The ~128 s utterance of BASELINE config 2 (T = 25,546 frames)."""
    return synth_utterance(tab, T_128S, utt_id)


def mixed_lengths(n: int, lo: int = 400, hi: int = T_128S, seed: int = 3) -> List[int]:
    """Config 3 lengths: T_b ~ round(U[lo, hi]) frames, seed-fixed."""
    rng = SplitMix64(SEED ^ (seed << 32))
    return [lo + rng.below(hi - lo + 1) for _ in range(n)]
