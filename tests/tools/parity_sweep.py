"""Parity sweep on the GPU box: N seeded utterances of random lengths, as ONE ragged batch through the HIP path and one
by one through the oracle (a process pool on the host cores): PCM relative RMS, excitation max |delta|, pulse positions
-- the worst of each, and the distribution.  The tests hold a handful of such cases each; this is the wide version.

    python tests/tools/parity_sweep.py [--n 240] [--max-frames 6000] [--procs 16] > profiles/rNN_parity_sweep.txt
"""
import argparse
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

DMAX = 1.7976931348623157e308


def oracle_one(args):
    """(voice metadata, utterance arrays) -> oracle PCM, excitation, pulses; runs in a worker process (no GPU)."""
    from oracle import oracle as O

    vmeta, durations, streams = args
    sts = []
    for (L, nwin, is_msd, use_gv, wlens, wcoef), (mean, var, msd, gvm, gvv, gsw, gvw, thr) in zip(vmeta["streams"], streams):
        msd = msd if msd is not None else np.full(len(durations), DMAX)
        sts.append(O.StreamStates(L, nwin, is_msd, use_gv, wlens, wcoef, mean, var, msd, gvm, gvv, gsw, gvw, thr))
    tr = [O.mlpg(s, durations) for s in sts]
    pcm, exc, pul = O.vocoder(vmeta["fs"], vmeta["fperiod"], vmeta["alpha"], 1.0, tr[1][:, 0], tr[0], tr[2], dumps=True)
    return pcm, exc, np.flatnonzero(pul != 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=240)
    ap.add_argument("--max-frames", type=int, default=6000)
    ap.add_argument("--procs", type=int, default=min(16, os.cpu_count() or 1))
    a = ap.parse_args()

    import jbonsai_amd as J
    from jbonsai_amd import synth
    from tests.conftest import VOICE
    from tests.helpers import EXC_TOL, PCM_TOL

    eng = J.Engine.load([VOICE])
    tab = synth.VoiceTables(eng)
    vi = eng.voice_info()
    rng = np.random.default_rng(20261005)
    lens = [int(x) for x in np.concatenate([rng.integers(1, 40, a.n // 6), rng.integers(40, 800, a.n // 2),
                                            rng.integers(800, a.max_frames, a.n - a.n // 6 - a.n // 2)])]
    utts = [synth.synth_utterance(tab, T, 7000 + k) for k, T in enumerate(lens)]
    vmeta = dict(fs=vi.sampling_frequency, fperiod=vi.fperiod, alpha=vi.alpha,
                 streams=[(s.vector_length, len(s.windows), s.is_msd, s.use_gv, [len(w) for w in s.windows],
                           [c for w in s.windows for c in w]) for s in vi.streams])
    jobs = [(vmeta, u.durations, [(s.mean, s.var, s.msd, s.gv_mean, s.gv_var, s.gv_switch, s.gv_weight, s.msd_threshold)
                                  for s in u.streams]) for u in utts]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(a.procs) as ex:
        fut = ex.map(oracle_one, jobs, chunksize=4)
        # the GPU batch beside the oracle's processes
        with J.Batch(vi, utts, keep_tracks=True) as b:
            b.run()
            b.sync()
            info = b.info()
            kern = b.kernel_info()
            got = [(b.pcm(i), b.excitation(i)) for i in range(len(utts))]
        refs = list(fut)
    t1 = time.perf_counter()
    rel, exc_abs, pulses, bad_pulse = [], [], 0, 0
    for (g, gx), (r, rx, rp), T in zip(got, refs, lens):
        assert len(g) == len(r) == T * vi.fperiod
        den = np.sqrt(np.mean(r * r))
        rel.append(float(np.sqrt(np.mean((g - r) ** 2)) / (den if den > 0 else 1.0)))
        exc_abs.append(float(np.abs(gx - rx).max()))
        # a pulse of the oracle is a sample where the excitation stands out by sqrt(pitch): positions are compared
        # through the excitation itself (its max |delta| above would be ~1 for a pulse a sample off)
        pulses += len(rp)
        bad_pulse += int(np.sum(np.abs(gx[rp] - rx[rp]) > 1e-6))
    rel, exc_abs = np.array(rel), np.array(exc_abs)
    print(f"{a.n} utterances of {min(lens)}..{max(lens)} frames ({sum(lens)} frames, {sum(lens) * vi.fperiod / vi.sampling_frequency:.0f} s of"
          f" audio) as one ragged batch: {kern[0]}, chunk {info['chunk_frames']} frames,"
          f" {info['n_items']} chunks, {info['n_redo']} redone; oracle on {a.procs} processes; {t1 - t0:.1f} s wall")
    print(f"PCM relative RMS vs oracle: max {rel.max():.3e}  median {np.median(rel):.3e}  (gate {PCM_TOL:g}; north_star 1e-4)")
    print(f"excitation max |delta|:     max {exc_abs.max():.3e}  median {np.median(exc_abs):.3e}  (gate {EXC_TOL:g})")
    print(f"pulses: {pulses} on the oracle's samples, {bad_pulse} off")
    worst = int(np.argmax(rel))
    print(f"worst utterance: #{worst}, {lens[worst]} frames, rel RMS {rel[worst]:.3e}")
    ok = rel.max() <= PCM_TOL and exc_abs.max() <= EXC_TOL and bad_pulse == 0
    print("PARITY GREEN" if ok else "PARITY RED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
