"""Shared test helpers: build state-level inputs with the ORACLE's front half and
hand the very same arrays to the HIP library."""
from __future__ import annotations

import numpy as np

import jbonsai_amd as J
from oracle import oracle as O

# ONE tolerance, stated once (VERDICT r5 "next" 5).  The vocoder runs an utterance as time-chunks and CERTIFIES every
# hand-off on the filter STATE: max|delta| <= VERIFY_TOL * max|state| (jb_batch_opts.verify_tol; the library's default:
# tests/test_abi.py holds this constant against jb_default_verify_tol() -- not asked here, at import, because loading
# the library while pytest COLLECTS would put its HIP runtime into the process before torch's, which later test modules
# import: with the two in that order hipOccupancyMaxActiveBlocksPerMultiprocessor answered 0 for every kernel).  The PCM behind a certified hand-off carries that difference while it decays; relative RMS of
# a whole utterance against the oracle, measured: worst 1.13e-9 over the 2,400 utterances of tests/tools/shape_sweep.py
# (orders 2-61, post-filter up to 0.5), 3.0e-10 over the 2,400 of parity_sweep.py, 3.4e-12 on BASELINE config 2's.
# The gate of every PCM comparison in tests/ and tests/tools/ is therefore 2 x VERIFY_TOL (north_star allows 1e-4).
# The excitation is feed-forward (no hand-off): its gate is absolute, on values of O(1..30).
VERIFY_TOL = 1e-9
PCM_TOL = 2.0 * VERIFY_TOL
EXC_TOL = 1e-9


def voice_info(v: "O.Voice", volume: float = 1.0, beta: float = 0.0) -> J.VoiceInfo:
    streams = [J.StreamInfo(v.vector_length[i], bool(v.is_msd[i]), bool(v.use_gv[i]), v.windows[i])
               for i in range(v.nstream)]
    return J.VoiceInfo(v.fs, v.fperiod, v.alpha, streams, volume=volume, beta=beta)


def oracle_states(v: "O.Voice", labels, speed=1.0, gv_weight=(1.0, 1.0, 1.0),
                  msd_threshold=(0.5, 0.5, 0.5)):
    dur = v.durations(labels, speed)
    sts = [v.stream_states(i, labels, gv_weight[i], msd_threshold[i]) for i in range(v.nstream)]
    return dur, sts


def to_utt(dur, sts) -> J.Utterance:
    out = []
    for s in sts:
        out.append(J.StreamStates(s.mean, s.var, s.msd if s.is_msd else None, s.gv_mean, s.gv_var,
                                  s.gv_switch, s.gv_weight, s.msd_threshold))
    return J.Utterance(dur, out)


def oracle_run(v: "O.Voice", dur, sts, volume=1.0, dumps=False, beta=0.0):
    """MLPG x3 + vocoder with the oracle on flat state arrays."""
    tracks = [O.mlpg(s, dur) for s in sts]
    r = O.vocoder(v.fs, v.fperiod, v.alpha, volume, tracks[1][:, 0], tracks[0], tracks[2], dumps=dumps,
                  beta=beta)
    return tracks, r


def rel_rms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b * b))
    return float(np.sqrt(np.mean((a - b) ** 2)) / (den if den > 0 else 1.0))
