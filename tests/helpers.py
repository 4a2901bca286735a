"""Shared test helpers: build state-level inputs with the ORACLE's front half and
hand the very same arrays to the HIP library."""
from __future__ import annotations

import numpy as np

import jbonsai_amd as J
from oracle import oracle as O


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
