"""jb_comm_* / jb_gather_pcm with MORE THAN ONE rank, executed before the driver's 8-GPU node does (VERDICT r3
"next" 5): RCCL refuses two ranks on one device, so the ranks bind the test double of
tests/fake_rccl (the nine entry points jb_multi.cpp uses, over shared memory) through JB_RCCL_LIBRARY.  Fresh
child processes, no torch, at most four on the device.  Covered: the ncclAllGather of the ragged slab lengths,
the grouped ncclSend / ncclRecv, root != 0, a rank with an empty slab, 16-bit slabs, and the collective failure
semantics -- a rank without a batch, and f64 mixed with 16-bit slabs: EVERY rank gets an error, none hangs."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def double():
    r = subprocess.run(["bash", str(ROOT / "tests" / "fake_rccl" / "build.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip().splitlines()[-1]


def run_world(tmp_path, double, world, root, mode):
    env = dict(os.environ, JB_RCCL_LIBRARY=double, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "tools" / "gather_worker.py"), str(tmp_path),
                               str(world), str(r), str(root), mode], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # the exact children started here
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{o}"
    return outs


@pytest.mark.parametrize("world,root,mode", [(4, 0, "f64"), (4, 2, "f64"), (3, 1, "i16"), (2, 0, "f64"),
                                             # an empty f64 batch on the root, 16-bit senders: not a mix,
                                             # and the receive slabs are sized by what the senders send
                                             (3, 1, "i16_f64_empty"), (3, 0, "i16_f64_empty")])
def test_gather_with_several_ranks_on_one_device(tmp_path, double, world, root, mode):
    outs = run_world(tmp_path, double, world, root, mode)
    assert all((tmp_path / f"done_{r}").read_text() == "ok" for r in range(world))
    assert f"(root): {world} slabs" in outs[root]


@pytest.mark.parametrize("world,root,mode", [(4, 0, "fail:3"), (3, 2, "fail:2"), (3, 0, "mixed")])
def test_a_failing_rank_fails_every_rank_and_hangs_none(tmp_path, double, world, root, mode):
    run_world(tmp_path, double, world, root, mode)
    assert all((tmp_path / f"done_{r}").read_text() == "error" for r in range(world))
