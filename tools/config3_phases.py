"""Where one pass of the config-3 job (bench.py --job config3, one GPU) spends its time: per sub-batch the
creation from pdf row indices (upload, device gather, work list), the GPU run up to the certified result,
and the release -- first strictly one after the other, then with the creation of the next sub-batch on a
helper thread as bench.py does it."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import shard, synth  # noqa: E402
from tests.conftest import VOICE  # noqa: E402

SUB = 7_000_000
eng = J.Engine.load([VOICE])
tab, vi = synth.VoiceTables(eng), eng.voice_info()
J.lib().jb_set_cached_memory_limit(160000)
lens = synth.mixed_lengths(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
k = max(1, -(-sum(lens) // SUB))
subs = [[synth.synth_utterance(tab, lens[i], 2000 + i, indexed=True) for i in sb] for sb in shard.lpt_partition(lens, k)]
pset = tab.pdf_set()
print(f"{len(lens)} utterances, {sum(lens)} frames, {k} sub-batches")
for rep in range(2):
    tot = dict(create=0.0, run=0.0, close=0.0)
    t_pass = time.perf_counter()
    for sb in subs:
        t0 = time.perf_counter()
        b = J.Batch(vi, sb, pdf_set=pset)
        t1 = time.perf_counter()
        b.run()
        b.sync()
        t2 = time.perf_counter()
        info, redo, tm = b.info(), b.redo_stats(), b.last_timing()
        b.close()
        t3 = time.perf_counter()
        tot["create"] += t1 - t0
        tot["run"] += t2 - t1
        tot["close"] += t3 - t2
        if rep == 1:
            print(f"  sub-batch {len(sb):4d} utts {sum(int(u.durations.sum()) for u in sb):8d} frames: create {1e3*(t1-t0):6.1f} ms, "
                  f"run+certify {1e3*(t2-t1):6.1f} ms (device {tm[0]:6.1f}, vocoder kernel {tm[1]:6.1f}), close {1e3*(t3-t2):5.1f} ms; "
                  f"chunk {info['chunk_frames']}, items {info['n_items']}, redone {info['n_redo']} ({redo[0]} at the checkpoint)")
    print(f"pass {rep} serial: {1e3*(time.perf_counter()-t_pass):.1f} ms = create {1e3*tot['create']:.1f} + run {1e3*tot['run']:.1f} + close {1e3*tot['close']:.1f}")
