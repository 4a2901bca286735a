"""PCM gather over RCCL (jb_comm_*, jb_gather_pcm): the optional sink of the utterance-sharded path that
wants every rank's PCM on one GPU.  One Comm per rank (process or thread per GPU); the
128-byte id made by rank 0 travels to the other ranks by the caller's own means (control plane)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import _ffi as F

ID_BYTES = 128


def unique_id() -> bytes:
    """A fresh communicator id (rank 0 calls this and hands it to every rank)."""
    buf = C.create_string_buffer(ID_BYTES)
    F.check(F.lib().jb_comm_unique_id(buf, ID_BYTES))
    return buf.raw


class Gathered:
    """What the root holds after a gather: one device slab per rank."""

    def __init__(self, handle, L, n_ranks, dtype):
        self._h, self._L, self.n_ranks, self.dtype = handle, L, n_ranks, dtype

    def samples(self, rank: int) -> int:
        return self._L.jb_gathered_samples(self._h, rank)

    def device_pointer(self, rank: int) -> Optional[int]:
        return self._L.jb_gathered_device(self._h, rank)

    def read(self, rank: int) -> np.ndarray:
        out = np.empty(self.samples(rank), dtype=self.dtype)
        F.check(self._L.jb_gathered_read(self._h, rank, out.ctypes.data, out.nbytes))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_gathered_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    def __init__(self, comm_id: Optional[bytes], n_ranks: int, rank: int, device: int = -1):
        L = F.lib()
        h = C.c_void_p()
        F.check(L.jb_comm_init(comm_id, n_ranks, rank, device, C.byref(h)))
        self._h, self._L, self.n_ranks, self.rank = h, L, n_ranks, rank

    def size(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount through jb_comm_size)."""
        return int(self._L.jb_comm_size(self._h))

    def gather_pcm(self, batch, root: int = 0):
        """Collective.  Returns (Gathered on the root / None elsewhere, milliseconds of the exchange)."""
        out, ms = C.c_void_p(), C.c_float()
        # batch = None: this rank's local work failed; it still joins, and every rank gets an error
        F.check(self._L.jb_gather_pcm(self._h, batch._h if batch is not None else None, root, C.byref(out),
                                      C.byref(ms)))
        if not out:
            return None, ms.value
        # the sample size is the SENDERS' (a root with an empty batch of the other kind takes theirs)
        dt = np.int16 if self._L.jb_gathered_sample_bytes(out) == 2 else np.float64
        return Gathered(out, self._L, self.n_ranks, dt), ms.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
