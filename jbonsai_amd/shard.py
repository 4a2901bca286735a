"""Multi-GPU sharding of a batch of independent utterances.

Utterances share only read-only tables, so the path shards with no data-path
collective: static LPT (longest-processing-time-first) partition by frame count so
that the sum of frames per GPU is balanced; one process per GPU."""
from __future__ import annotations

import heapq
from typing import List, Sequence


def lpt_partition(lengths: Sequence[int], n_parts: int) -> List[List[int]]:
    """Indices of `lengths` assigned to each of n_parts bins, longest first onto the
    currently lightest bin.  Deterministic; every index appears exactly once."""
    if n_parts <= 0:
        raise ValueError("n_parts must be positive")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    heap = [(0, p) for p in range(n_parts)]
    heapq.heapify(heap)
    parts: List[List[int]] = [[] for _ in range(n_parts)]
    for i in order:
        load, p = heapq.heappop(heap)
        parts[p].append(i)
        heapq.heappush(heap, (load + int(lengths[i]), p))
    return parts


def lpt_partition_native(lengths: Sequence[int], n_parts: int) -> List[List[int]]:
    """The same partition from the library (jb_lpt_partition, the rule the multi-device entries
    jb_synthesize_batch_multi / jb_paramgen_vocode_batch_multi split by); needs no GPU."""
    import ctypes as C

    from . import _ffi as F

    n = len(lengths)
    w = (C.c_uint64 * max(1, n))(*[int(x) for x in lengths])
    part = (C.c_uint32 * max(1, n))()
    F.check(F.lib().jb_lpt_partition(w, n, n_parts, part))
    parts: List[List[int]] = [[] for _ in range(n_parts)]
    # bins list their items heaviest first, as lpt_partition does
    for i in sorted(range(n), key=lambda i: (-int(lengths[i]), i)):
        parts[part[i]].append(i)
    return parts


def shard_for_rank(lengths: Sequence[int], rank: int, world: int) -> List[int]:
    return lpt_partition(lengths, world)[rank]


def imbalance(lengths: Sequence[int], parts: List[List[int]]) -> float:
    loads = [sum(int(lengths[i]) for i in p) for p in parts]
    mean = sum(loads) / max(1, len(loads))
    return (max(loads) / mean) if mean > 0 else 1.0
