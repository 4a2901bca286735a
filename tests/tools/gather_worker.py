"""One rank of the N > 1 rehearsal of jb_comm_* / jb_gather_pcm on ONE device (tests/test_gpu_gather_world.py):
a fresh process, no torch, RCCL = the test double (JB_RCCL_LIBRARY).  Rank 0 makes the communicator id and
leaves it in the exchange directory; every rank runs a small batch of its own, joins the gather, and the root
checks every slab against what the ranks left behind.
usage: gather_worker.py DIR WORLD RANK ROOT MODE      MODE = f64 | i16 | fail:<rank> | mixed | i16_f64_empty
(i16_f64_empty: every rank with samples makes 16-bit PCM, rank 1 -- whose batch is empty -- an f64 batch: not a
mix, and the slabs are sized by the senders, whichever rank is the root)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import jbonsai_amd as J  # noqa: E402
from jbonsai_amd import synth  # noqa: E402

d, world, rank, root, mode = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
VOICE = ROOT / "tests" / "golden" / "voice" / "nitech_jp_atr503_m001.htsvoice"


def wait_for(p, timeout=120.0):
    t0 = time.time()
    while not p.exists():
        if time.time() - t0 > timeout:
            raise SystemExit(f"rank {rank}: {p} never appeared")
        time.sleep(0.02)


if rank == 0:
    (d / "id.tmp").write_bytes(J.comm.unique_id())
    (d / "id.tmp").rename(d / "id.bin")
wait_for(d / "id.bin")
cid = (d / "id.bin").read_bytes()

eng = J.Engine.load([VOICE])
tab, vi = synth.VoiceTables(eng), eng.voice_info()
# ragged shares: rank r holds r + 1 utterances of different lengths; rank 1 holds none (an empty slab)
n_utts = 0 if rank == 1 else rank + 1
utts = [synth.synth_utterance(tab, 150 + 90 * ((rank * 7 + i) % 5), 900 + 10 * rank + i) for i in range(n_utts)]
i16 = mode == "i16" or (mode == "mixed" and rank == world - 1) or (mode == "i16_f64_empty" and rank != 1)
batch = J.Batch(vi, utts, pcm_i16=i16)
batch.run()
batch.sync()
mine = [batch.pcm_i16(i) if i16 else batch.pcm(i) for i in range(n_utts)]
slab = np.concatenate(mine) if mine else np.zeros(0, dtype=np.int16 if i16 else np.float64)
assert [batch.pcm_offset(i) for i in range(n_utts)] == list(np.cumsum([0] + [len(m) for m in mine])[:-1])
np.save(d / f"slab_{rank}.npy", slab)

comm = J.comm.Comm(cid, world, rank)
fail_rank = int(mode.split(":")[1]) if mode.startswith("fail:") else -1
expect_error = fail_rank >= 0 or mode == "mixed"
try:
    g, ms = comm.gather_pcm(None if rank == fail_rank else batch, root=root)
except J.JbError as e:
    if not expect_error:
        raise
    print(f"rank {rank}: error as expected: {e}")
    (d / f"done_{rank}").write_text("error")
    sys.exit(0)
if expect_error:
    raise SystemExit(f"rank {rank}: the gather succeeded where every rank should have got an error")
if rank == root:
    assert g is not None
    for p in range(world):
        wait_for(d / f"slab_{p}.npy")
        want = np.load(d / f"slab_{p}.npy")
        assert g.samples(p) == len(want), (p, g.samples(p), len(want))
        got = g.read(p)
        assert got.dtype == want.dtype or len(want) == 0, (p, got.dtype, want.dtype)
        assert np.array_equal(got, want), p
    print(f"rank {rank} (root): {world} slabs, {[g.samples(p) for p in range(world)]} samples, {ms:.2f} ms")
    g.close()
else:
    assert g is None
(d / f"done_{rank}").write_text("ok")
comm.close()
batch.close()
