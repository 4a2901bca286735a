"""How far does the band solve of MLPG (mlpg.rs:25-142, no GV) see?  The track of frames [s, e) from the solve of the
WHOLE utterance against the track of the same frames from the solve of the window [s - H, e + H) alone (states cut at
the window's ends), for growing halo H.  If a modest halo reproduces the whole-utterance track to the last bits, the band
solve can be cut in time like the vocoder -- the question DESIGN.md section 8 asks before the build / band solve / GV
chain is fused into one resident kernel.  CPU only (the checker library); prints a table.

    python tests/tools/mlpg_halo_study.py [frames] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from tests.conftest import VOICE  # noqa: E402

DMAX = 1.7976931348623157e308


def cut(dur, lo, hi):
    """states and durations of frames [lo, hi)"""
    ends = np.cumsum(dur)
    starts = ends - dur
    keep = np.nonzero((ends > lo) & (starts < hi))[0]
    d = (np.minimum(ends[keep], hi) - np.maximum(starts[keep], lo)).astype(np.uint32)
    return keep, d


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    v = O.Voice(VOICE)
    rng = np.random.default_rng(seed)
    # a synthetic state sequence from the voice's real leaves, like jbonsai_amd.synth (stream 0 = MCP)
    si = 0
    L = v.vector_length[si]
    tabs = [v.pdf_table(1 + si, t) for t in range(v.ntree(1 + si))]
    S = T // 5
    rows = np.stack([tabs[k % len(tabs)][rng.integers(len(tabs[k % len(tabs)]))] for k in range(S)])
    WL = 3 * L
    mean, var = rows[:, :WL].astype(np.float64), rows[:, WL:2 * WL].astype(np.float64)
    dur = rng.integers(2, 9, S).astype(np.uint32)
    T = int(dur.sum())
    win = v.windows[si]
    wl = [len(w) for w in win]
    wc = [c for w in win for c in w]

    def solve(keep, d):
        st = O.StreamStates(L, 3, False, False, wl, wc, mean[keep], var[keep], np.full(len(keep), DMAX), None, None, None,
                            1.0, 0.5)
        return O.mlpg(st, d)

    full = solve(np.arange(S), dur)
    s, e = T // 2 - 244, T // 2 + 244
    ref = full[s:e]
    scale = np.abs(ref).max(axis=0)
    print(f"utterance of {T} frames, window [{s}, {e}), {L} dims; max |window - whole| / max |track| per halo:")
    for H in (0, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192):
        lo, hi = max(0, s - H), min(T, e + H)
        keep, d = cut(dur, lo, hi)
        got = solve(keep, d)[s - lo:s - lo + (e - s)]
        err = (np.abs(got - ref).max(axis=0) / scale)
        exact = int(np.sum(np.all(got == ref, axis=0)))
        print(f"  H = {H:4d}: worst dim {err.max():.3e}, median dim {np.median(err):.3e}, dims bit-identical {exact}/{L}")


if __name__ == "__main__":
    main()
