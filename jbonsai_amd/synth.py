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
    """pdf tables of one voice of the engine (default: voice 0), pulled through the C ABI
    (jb_engine_pdf_table)."""

    def __init__(self, eng: Engine, voice: int = 0):
        self.eng = eng
        self.vi = eng.voice_info()
        self.nstate = eng.num_states
        self.dur = eng.pdf_table(0, 0, voice)
        self.stream = []
        self.gv = []
        for s in range(eng.num_streams):
            nt, _ = eng.model_shape(1 + s, voice)
            self.stream.append([eng.pdf_table(1 + s, t, voice) for t in range(nt)])
            if self.vi.streams[s].use_gv:
                self.gv.append(eng.pdf_table(4 + s, 0, voice))
            else:
                self.gv.append(None)
        # all trees of a stream concatenated (the layout of jb_pdf_table) and each tree's first row
        self.concat = [np.concatenate(t, axis=0) for t in self.stream]
        self.tree_off = [np.cumsum([0] + [len(x) for x in t[:-1]]) for t in self.stream]

    def pdf_set(self, device: int = -1) -> PdfSet:
        """The voice's stream tables resident on `device` (for indexed batches)."""
        return PdfSet([self.concat], device=device)


_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _splitmix_block(seed: int, first: int, count: int) -> np.ndarray:
    """Outputs first .. first+count-1 of SplitMix64(seed) at once (it is a counter-based generator:
    output k mixes seed + (k+1)*GAMMA)."""
    with np.errstate(over="ignore"):
        k = np.arange(first + 1, first + count + 1, dtype=np.uint64)
        z = np.uint64(seed & MASK) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _weighted(rows, w):
    """VoiceSet::weighted (voice_set.rs:80-95): first*w0, then += w_i*row_i in voice order, in f64."""
    acc = np.asarray(rows[0], dtype=np.float64) * w[0]
    for r, wi in zip(rows[1:], w[1:]):
        acc += wi * np.asarray(r, dtype=np.float64)
    return acc


def _draw_states(tabs, wd, target_frames: int, utt_id: int):
    """The random part of an utterance: per phone one duration pdf and one leaf per (stream, state) --
    16 draws per phone with the nitech shapes, in that order --, phones until target_frames is reached,
    the last state trimmed.  Returns (durations [S], rows per stream [S], S)."""
    tab = tabs[0]
    ns, nstream = tab.nstate, len(tab.stream)
    per = 1 + nstream * ns
    seed = SEED ^ utt_id
    # leaves are drawn modulo the size of the (stream, state) tree
    mod = np.asarray([[len(tab.stream[s][min(k, len(tab.stream[s]) - 1)]) for k in range(ns)]
                      for s in range(nstream)], dtype=np.uint64)
    off = np.asarray([[int(tab.tree_off[s][min(k, len(tab.stream[s]) - 1)]) for k in range(ns)]
                      for s in range(nstream)], dtype=np.int64)
    ndur = np.uint64(len(tab.dur))
    durs, rows, total, phones = [], [], 0, 0
    while total < target_frames:
        n = max(16, (target_frames - total) // 12 + 8)
        z = _splitmix_block(seed, phones * per, n * per).reshape(n, per)
        di = (z[:, 0] % ndur).astype(np.int64)
        dp = tab.dur[di][:, :ns].astype(np.float64) if wd is None else _weighted([t.dur[di][:, :ns] for t in tabs], wd)
        d = np.maximum(1.0, np.floor(dp + 0.5)).astype(np.int64)              # [n, ns]
        leaf = (z[:, 1:].reshape(n, nstream, ns) % mod[None]).astype(np.int64) + off[None]
        durs.append(d)
        rows.append(leaf)
        total += int(d.sum())
        phones += n
    d = np.concatenate(durs).reshape(-1)                                       # state-major: phone, state
    leaf = np.concatenate(rows)                                                # [phones, nstream, ns]
    cs = np.cumsum(d)
    S = int(np.searchsorted(cs, target_frames, side="left")) + 1               # first state reaching the target
    d = d[:S].copy()
    d[S - 1] -= int(cs[S - 1]) - target_frames
    ridx = [leaf[:, s, :].reshape(-1)[:S].astype(np.uint32) for s in range(nstream)]
    return d.astype(np.uint32), ridx, S


def _assemble(tabs, weights, durs, ridx, S, indexed):
    tab = tabs[0]
    ns, nstream, nv = tab.nstate, len(tab.stream), len(tabs)
    nphone = (S + ns - 1) // ns
    sw = np.ones(S, dtype=np.uint8)
    sw[:ns] = 0                      # first / last phone behave like silence: GV off
    sw[(nphone - 1) * ns:] = 0
    streams = []
    for s in range(nstream):
        si = tab.vi.streams[s]
        WL = si.vector_length * len(si.windows)
        wp = np.ones(1) if weights is None else np.asarray(weights["parameter"][s], dtype=np.float64)
        gm = gv = gs = None
        if si.use_gv:
            g = (tab.gv[s][0].astype(np.float64) if weights is None else
                 _weighted([t.gv[s][0] for t in tabs], np.asarray(weights["gv"][s], dtype=np.float64)))
            gm, gv, gs = g[:si.vector_length], g[si.vector_length:], sw
        if indexed:
            streams.append(IndexStreamStates([ridx[s]] * nv, wp, gm, gv, gs, 1.0, 0.5))
            continue
        arr = (tab.concat[s][ridx[s]].astype(np.float64) if weights is None else
               _weighted([t.concat[s][ridx[s]] for t in tabs], wp))
        msd = arr[:, 2 * WL] if si.is_msd else None
        streams.append(StreamStates(arr[:, :WL], arr[:, WL:2 * WL], msd, gm, gv, gs, 1.0, 0.5))
    return IndexUtterance(durs, streams) if indexed else Utterance(durs, streams)


def synth_utterance(tab: VoiceTables, target_frames: int, utt_id: int = 0, indexed: bool = False):
    """indexed=False: state-level Utterance (per-state Gaussians expanded on the host);
    indexed=True: the SAME utterance as pdf row indices (IndexUtterance) for a batch created over
    VoiceTables.pdf_set() -- 12 bytes per state cross PCIe instead of 2.2 kB."""
    durs, ridx, S = _draw_states([tab], None, target_frames, utt_id)
    return _assemble([tab], None, durs, ridx, S, indexed)


def voice_set_pdf_set(tabs: List[VoiceTables], device: int = -1) -> PdfSet:
    """The stream tables of several voices resident on `device` (BASELINE config 5)."""
    return PdfSet([t.concat for t in tabs], device=device)


def synth_utterance_voices(tabs: List[VoiceTables], weights: dict, target_frames: int, utt_id: int = 0,
                           indexed: bool = False):
    """synth_utterance over SEVERAL voices (BASELINE config 5: two-voice interpolation).  The same
    leaves are drawn as for one voice (the voices of a set share their trees' shapes); every voice
    contributes ITS row of that leaf, blended with
    weights = {"duration": [nv], "parameter": [[nv] per stream], "gv": [[nv] per stream]}.
    indexed=False: the blend is formed here in the reference's order (voice_set.rs:80-95; what a checker
    is given); indexed=True: row indices per voice + weights, blended on the device
    (k_gather_blend)."""
    durs, ridx, S = _draw_states(tabs, np.asarray(weights["duration"], dtype=np.float64), target_frames, utt_id)
    return _assemble(tabs, weights, durs, ridx, S, indexed)


def u128(tab: VoiceTables, utt_id: int = 0) -> Utterance:
    """The ~128 s utterance of BASELINE config 2 (T = 25,546 frames)."""
    return synth_utterance(tab, T_128S, utt_id)


def mixed_lengths(n: int, lo: int = 400, hi: int = T_128S, seed: int = 3) -> List[int]:
    """Config 3 lengths: T_b ~ round(U[lo, hi]) frames, seed-fixed."""
    rng = SplitMix64(SEED ^ (seed << 32))
    return [lo + rng.below(hi - lo + 1) for _ in range(n)]


def with_order(vi, u: Utterance, nmcp: int, seed: int = 7):
    """(voice info, utterance) of ANOTHER mel-cepstral order (vector length `nmcp` = order + 1): the MCP stream of a
    state-level utterance cut to `nmcp` dimensions per window, or widened by small seeded dimensions -- the
    reference is generic in the order (vocoder/mod.rs:45-70, mlsa.rs:38-45), the voices at hand are not (nitech: 35).
    What tests and bench.py's `order49` record run the other instantiations of the throughput kernel on."""
    import dataclasses

    rng = np.random.default_rng(seed)
    L, W = vi.streams[0].vector_length, len(vi.streams[0].windows)
    m0, S = u.streams[0], len(u.durations)

    def reshape(a, fill):
        a = np.asarray(a).reshape(S, W, L)
        if nmcp <= L:
            return a[:, :, :nmcp].reshape(S, W * nmcp).copy()
        return np.concatenate([a, fill((S, W, nmcp - L))], axis=2).reshape(S, W * nmcp)

    mean = reshape(m0.mean, lambda sh: 0.02 * rng.standard_normal(sh))
    var = reshape(m0.var, lambda sh: 0.01 + 0.01 * rng.random(sh))
    gvm = m0.gv_mean[:nmcp] if nmcp <= L else np.concatenate([m0.gv_mean, np.full(nmcp - L, 4e-4)])
    gvv = m0.gv_var[:nmcp] if nmcp <= L else np.concatenate([m0.gv_var, np.full(nmcp - L, 1e-8)])
    s0 = dataclasses.replace(m0, mean=mean, var=var, gv_mean=gvm, gv_var=gvv)
    st0 = dataclasses.replace(vi.streams[0], vector_length=nmcp)
    return (dataclasses.replace(vi, streams=[st0] + list(vi.streams[1:])),
            Utterance(u.durations, [s0] + list(u.streams[1:])))
