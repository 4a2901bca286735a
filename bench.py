#!/usr/bin/env python3
"""bench.py -- 48 kHz PCM samples/s of the parameter-generation + MLSA-vocoder hot
path on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one pass of the whole hot path (MLPG+GV x3 -> frame prologue -> pulse
schedule -> excitation + MLSA cascade) over one batch of synthetic utterances
whose state-level inputs are already resident in HBM.

Workloads (--job):
  config2 (default, the headline): BASELINE config 2 per GPU -- a batch of 256 copies of the
      ~128 s utterance (T = 25,546 frames, 6,131,040 samples each) built from real nitech pdfs
      (jbonsai_amd/synth.py).  N > 1: every rank runs the per-GPU workload (weak scaling).
  config3: BASELINE config 3 as ONE job -- the seed-fixed list of 4096 mixed-length utterances
      (synth.mixed_lengths) is built once, LPT-partitioned over the N ranks (jbonsai_amd/shard.py)
      and every rank synthesises its share in sub-batches that fit HBM (strong scaling: total work
      fixed).  At N > 1 a short config-3 pass is also appended to the default run and reported
      under "config3_strong" (outside `value`).
Utterances are independent, so there is no data-path collective; RCCL carries the barrier, the
max-over-ranks of the timed region and -- with --gather -- the PCM gather onto rank 0.

Launch: `python bench.py --gpus N` spawns its N ranks itself (fresh child processes, started
before this process touches HIP or torch.cuda); under torch.distributed.run (WORLD_SIZE set) the
process is one rank.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
# more hardware queues than HIP's default 4, before anything initialises HIP (torch may come first): batches in
# flight overlap only on queues of their own
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, str(ROOT))

B_ALG = 8.67          # algorithmic bytes per output sample, f64 PCM
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
PATH_FLOP_PER_SAMPLE = 1.39e3  # algorithmic f64 flops per output sample of the WHOLE path (vocoder + MLPG/GV)
FLOP_PER_SAMPLE = 1.35e3      # ... of what the dominant kernel computes (V5-V9: gain, df1, df2 + fir, interpolation;
                              # SURVEY.md 8a "Vocoder total"): the figure the kernel's roofline is priced with
# FP64 vector peak = half the guide's 157.3 TFLOPS FP32 vector rate (1024 SIMDs x 16 lanes x 2 flop x
# 2.4 GHz); tools/microbench/f64_rate.hip sustains 68 TFLOP/s of dependent-free v_fma_f64 on this part
FP64_VALU_PEAK_TFLOPS = 78.6
VOICE = ROOT / "tests" / "golden" / "voice" / "nitech_jp_atr503_m001.htsvoice"
CONFIG3_UTTS = 4096           # BASELINE config 3: "Batch=4096 mixed-length synthetic ... sequences"
SUB_BATCH_FRAMES = 7_000_000  # frames per sub-batch of the config-3 job (config 2 is 6.54 M: ~62 GB of HBM)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--job", choices=("config2", "config3"), default="config2",
                    help="config2: the headline (per-GPU batch of copies, weak scaling); config3: the 4096 "
                         "mixed-length utterances as one LPT-sharded job (strong scaling)")
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU (config2 job)")
    ap.add_argument("--frames", type=int, default=0, help="frames per utterance (0 = 25,546 = ~128 s)")
    ap.add_argument("--utts", type=int, default=CONFIG3_UTTS, help="utterances of the config3 job (whole job)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements appended to the default line (distinct utterances, "
                         "D2H-inclusive, labels -> PCM, configs 4 and 5 at batch 1024, the config-3 job); none of "
                         "them enters `value`")
    ap.add_argument("--distinct", type=int, default=1,
                    help="number of DISTINCT synthetic utterances tiled over the batch (default 1 = BASELINE "
                         "config 2's copies of one utterance); >1 shows the cost of real hand-off failures")
    ap.add_argument("--seed", type=int, default=0,
                    help="id of the synthetic utterance the batch holds copies of (default 0 = the headline's; "
                         "the extras report ids 0-3 as `copies_over_seeds`)")
    ap.add_argument("--mixed", action="store_true",
                    help="config2 job on BASELINE config 3's per-GPU share instead: --batch distinct utterances "
                         "of seed-fixed lengths U[400, 25546] frames, resident before the timed region")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed steps, gather every rank's PCM slab to rank 0 with RCCL "
                         "(point-to-point over xGMI): the 16-bit slab (a quarter of the bytes: at N = 8 the f64 "
                         "slabs are 88 GB into the root per step, more than its seven links carry in a step), alone "
                         "(gather_i16_ms / gather_f64_ms) and beside the next step on two alternating batches "
                         "(gather_overlapped_ms_per_step); never part of `value` (SURVEY 8e)")
    ap.add_argument("--gather-f64", action="store_true", help="gather the f64 slabs instead (link-bound at N = 8)")
    ap.add_argument("--beta", type=float, default=0.0,
                    help="post-filter coefficient (off-config: BASELINE's metric is quoted at beta = 0)")
    ap.add_argument("--kernel", choices=("auto", "wave", "triple"), default="auto",
                    help="vocoder kernel of the timed batch (auto = the library's choice by batch size; measurement aid)")
    ap.add_argument("--chunk-frames", type=int, default=0, help="chunk length of the timed batch (0 = the library's choice)")
    ap.add_argument("--warmup-frames", type=int, default=0, help="warm-up of a chunk in frames (0 = the library's choice)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="batches in flight per GPU (2: one batch's parameter generation overlaps the "
                         "other's vocoder on separate HIP streams; 1: strictly one step at a time)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without a torchrun environment spawns its own ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """N fresh child processes, one per GPU, started BEFORE this process makes any HIP / torch.cuda
    call (it never does).  Rank 0 prints the JSON line on the inherited stdout.
    The launcher never waits for ever and never leaves its ranks behind: JB_BENCH_TIMEOUT_S (default 1500)
    bounds the whole run, SIGTERM / SIGINT end exactly the children it started (terminate, then kill) and
    the launcher exits non-zero."""
    import signal

    port = _free_port()
    procs = []
    deadline = time.monotonic() + float(os.environ.get("JB_BENCH_TIMEOUT_S", "1500"))

    def stop_children(grace=5.0):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + grace
        for p in procs:
            while p.poll() is None and time.monotonic() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()

    class _Stop(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Stop(signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    # a failed rank leaves the others waiting at a collective: end exactly the
                    # children this launcher started
                    for q in pending:
                        q.terminate()
            if pending and time.monotonic() > deadline:
                print(f"bench.py: ranks still running at the deadline (JB_BENCH_TIMEOUT_S); ending them", file=sys.stderr)
                rc = rc or 124
                break
            time.sleep(0.05)
    except _Stop as st:
        rc = 128 + int(st.args[0])
    finally:
        stop_children()
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


# ---------------------------------------------------------------------------------------------
def _host_cpu():
    """CPU model string, nproc, affinity and cgroup quota of this box."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    nproc = os.cpu_count() or 1
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    quota = None
    try:  # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return model, nproc, aff, quota


def cpu_baseline(utt, vi, batch_size, gpu=None):
    """the oracle (C restatement of jbonsai's CPU path, kind="port") built
    -O3 -march=native ON THIS BOX (oracle/Makefile `native`; no FMA contraction, no fast-math: same bits
    as the checker build), on the same synthetic utterance as the GPU batch:
      (a) single thread: one utterance, median of 5 runs after 1 warm-up;
      (b) all host cores: one utterance per thread at a time until the batch is done or ~12 s have
          passed (a bounded sample of the batch: the default run must finish within minutes)."""
    import numpy as np
    from oracle import oracle as O

    model, nproc, aff, quota = _host_cpu()
    build = "gcc -O3 -march=native -ffp-contract=off (oracle/Makefile native, built on this box)"
    try:
        O.use_library(O.build_native())
    except Exception as e:  # the checker build still measures something
        O.use_library(None)
        build = f"gcc -O2 -ffp-contract=off (native build failed: {e!r})"
    threads = max(1, int(os.environ.get("JB_CPU_THREADS", "0")) or min(aff, int(quota + 0.5) if quota else aff))
    sts = []
    for i, s in enumerate(utt.streams):
        si = vi.streams[i]
        msd = s.msd if s.msd is not None else np.full(len(utt.durations), 1.7976931348623157e308)
        sts.append(O.StreamStates(si.vector_length, len(si.windows), si.is_msd, si.use_gv,
                                  [len(w) for w in si.windows], [c for w in si.windows for c in w],
                                  s.mean, s.var, msd, s.gv_mean, s.gv_var, s.gv_switch))
    nsamp = int(utt.durations.sum()) * vi.fperiod

    def one(keep=False):
        tr = [O.mlpg(s, utt.durations) for s in sts]
        return O.vocoder(vi.sampling_frequency, vi.fperiod, vi.alpha, 1.0, tr[1][:, 0], tr[0], tr[2], dumps=keep)

    # warm-up -- and the error figure of the metric (SURVEY 8d "Error metric"): the oracle's PCM and excitation of
    # the batch's utterance against what the GPU produced for it inside the timed batch (utterance 0 and the
    # last one) and in a one-utterance batch with the excitation tap
    ref_pcm, ref_exc, ref_pul = one(keep=True)
    err = None
    if gpu is not None:
        def rel(a):
            return float(np.sqrt(np.mean((a - ref_pcm) ** 2)) / np.sqrt(np.mean(ref_pcm ** 2)))
        same_len = all(len(a) == len(ref_pcm) for a in gpu["pcm"])
        err = {"length_equal": bool(same_len), "samples": int(len(ref_pcm)), "utterances_compared": len(gpu["pcm"]),
               "gate": {"rel_rms": 1e-9, "north_star_rms": 1e-4}}
        if same_len:
            err["rel_rms"] = max(rel(a) for a in gpu["pcm"])
            err["max_abs_i16_scale"] = max(float(np.max(np.abs(a - ref_pcm))) for a in gpu["pcm"])
            err["ref_rms_i16_scale"] = float(np.sqrt(np.mean(ref_pcm ** 2)))
            if gpu.get("exc") is not None and len(gpu["exc"]) == len(ref_exc):
                d = float(np.max(np.abs(gpu["exc"] - ref_exc)))
                err["excitation_max_abs"] = d
                err["pulses"] = int(np.count_nonzero(ref_pul))
                # a pulse one sample off is an excitation error of the pulse's size (sqrt(period) * tap ~ 1..5)
                err["pulse_positions_equal"] = bool(d < 1e-6)
    del ref_pcm, ref_exc, ref_pul
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        one()
        runs.append(time.perf_counter() - t0)
    t_single = sorted(runs)[len(runs) // 2]
    budget_s = float(os.environ.get("JB_CPU_BASELINE_S", "12"))
    lock, done = threading.Lock(), [0]
    t0 = time.perf_counter()

    def work():
        while True:
            with lock:
                if done[0] >= batch_size or time.perf_counter() - t0 > budget_s:
                    return
                done[0] += 1
            one()

    th = [threading.Thread(target=work) for _ in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    # the reference's own benchmark shapes (benches/bonsais.rs): one sentence per call, labels in, PCM out, one thread
    single = {}
    try:
        from tests.golden.labels import BENCH_LETTER, SAMPLE_SENTENCE_1, SAMPLE_SENTENCE_2

        ov = O.Voice(VOICE)
        for name, lab in (("bonsai_8_labels", SAMPLE_SENTENCE_1), ("is_bonsai_20_labels", SAMPLE_SENTENCE_2),
                          ("bonsai_letter_43_labels", BENCH_LETTER)):
            ov.synthesize(lab)
            ts = []
            for _ in range(7):
                t1 = time.perf_counter()
                ov.synthesize(lab)
                ts.append((time.perf_counter() - t1) * 1e3)
            single[name] = sorted(ts)[len(ts) // 2]
    except Exception as e:  # a secondary figure
        single = {"error": repr(e)}
    O.use_library(None)
    all_cores = nsamp * done[0] / dt
    return {
        "value": all_cores, "unit": "samples/s", "cores": threads, "kind": "port",
        "single_thread": {"value": nsamp / t_single, "unit": "samples/s", "seconds_per_utterance": t_single,
                          "realtime_factor": nsamp / t_single / vi.sampling_frequency,
                          "runs": 5, "statistic": "median after 1 warm-up"},
        "all_cores": {"value": all_cores, "unit": "samples/s", "threads": threads, "utterances": done[0],
                      "of_batch": batch_size, "wall_s": dt,
                      "realtime_factor": all_cores / vi.sampling_frequency},
        "error_vs_oracle": err, "single_sentence_ms": single,
        "cpu_model": model, "nproc": nproc, "affinity": aff, "cgroup_cpu_quota": quota, "build": build,
        "sample": f"{done[0]} of the batch's {batch_size} utterances of {nsamp} samples (the same synthetic utterance "
                  f"as the GPU batch), one per thread at a time on {threads} threads, {dt:.2f} s wall; single thread: "
                  f"median of 5 runs; C restatement of jbonsai's CPU path (oracle/), never jbonsai itself "
                  "(no Rust toolchain).  Anchor: jbonsai reaches ~7.6 Msamples/s per i5-13500 core (README.md:84)",
    }


def pcm_slab_tensor(batch):
    """The batch's PCM slab (jb_batch_device_pcm) as a zero-copy torch tensor on its device (f64,
    or i16 for a pcm_i16 batch), for an RCCL gather by the caller.  The tensor aliases
    library-owned memory: valid until the batch is closed.  Lives here, not in the package: the
    product does not import torch."""
    import torch

    p, n = batch.device_pcm()
    i16 = bool(batch.flags & 64)  # JB_BATCH_PCM_I16

    class _Slab:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i2" if i16 else "<f8",
                                    "data": (int(p), False), "version": 2}

    dev = batch.device if batch.device >= 0 else torch.cuda.current_device()
    return torch.as_tensor(_Slab(), device=torch.device("cuda", dev))


class Ranks:
    """Rendezvous of the N ranks (torch.distributed; backend nccl = RCCL, or gloo on host tensors in
    the one-GPU rehearsal mode JB_BENCH_REHEARSE=1 where every rank uses device 0)."""

    def __init__(self):
        import torch

        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # JB_BENCH_DRYRUN=1 (CPU test of the launcher and the rank plumbing, tests/test_shard_dist.py):
        # rendezvous over gloo, the config-3 plan is built, no batch is created and nothing is measured
        self.dry = os.environ.get("JB_BENCH_DRYRUN", "0") != "0"
        self.rehearse = self.dry or os.environ.get("JB_BENCH_REHEARSE", "0") != "0"
        self.dist = None
        if self.rehearse:
            self.local_rank = 0
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.dist = dist
            if not self.dry:
                torch.cuda.set_device(self.local_rank)
            if self.rehearse:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
        if not self.dry:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
            torch.cuda.set_device(self.local_rank)
        self.tdev = "cpu" if self.rehearse else "cuda"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize()

    def max(self, x: float) -> float:
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_floats(self, xs):
        """[world][len(xs)] of every rank's list."""
        if self.dist is None:
            return [list(xs)]
        t = self.torch.tensor(list(xs), dtype=self.torch.float64, device=self.tdev)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[float(v) for v in o.tolist()] for o in out]

    def gather_slabs(self, slab):
        """Every rank's PCM slab onto rank 0 (north_star's optional sink).  The slabs differ in length
        (config 3 shares), so this is point-to-point: rank 0 posts one receive per peer, every peer
        one send -- over xGMI each peer uses its own link into rank 0.  Returns the time in ms."""
        if self.dist is None:
            return None
        torch, dist = self.torch, self.dist
        if self.rehearse:
            slab = slab.cpu()
        n = torch.tensor([slab.numel()], dtype=torch.int64, device=self.tdev)
        sizes = [torch.empty_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        bufs = None
        if self.rank == 0:
            bufs = [torch.empty(int(s.item()), dtype=slab.dtype, device=slab.device) for s in sizes[1:]]
        self.barrier()
        t0 = time.perf_counter()
        if self.rank == 0:
            ops = [dist.P2POp(dist.irecv, b, r + 1) for r, b in enumerate(bufs) if b.numel()]
        else:
            ops = [dist.P2POp(dist.isend, slab, 0)] if slab.numel() else []
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.barrier()
        ms = (time.perf_counter() - t0) * 1e3
        del bufs
        return ms

    def gather_native(self, J, batch, pair=None):
        """jb_gather_pcm: every rank's PCM slab onto rank 0 through the library's RCCL binding (no torch
        tensor involved; torch.distributed only hands the 128-byte communicator id to the ranks).  Returns
        (max-over-ranks time of one exchange alone in ms, ms per step with the gather of step k beside step k + 1 on
        the two alternating batches of `pair` or None); the communicator setup is not part of either."""
        import threading

        ids = [J.comm.unique_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(ids, src=0)
        c = J.comm.Comm(ids[0], self.world, self.rank, device=self.local_rank)
        self.gather_comm_size = c.size()  # ncclCommCount of the communicator the library formed for the gather
        self.barrier()
        g, ms = c.gather_pcm(batch, root=0)
        ms = self.max(ms)
        if g is not None:
            assert sum(g.samples(r) for r in range(self.world)) >= batch.total_samples
            g.close()
        overlapped = None
        if pair is not None:
            # step k + 1 on one batch while the slab of step k (the other batch) travels: the gather runs on the
            # communicator's own stream from a thread of its own (jb_gather_pcm waits for ITS batch only); every rank
            # issues its gathers in the same order, so the collective matches up
            errs = []

            def gather(j):
                try:
                    gj, _ = c.gather_pcm(pair[j], root=0)
                    if gj is not None:
                        gj.close()
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))

            for b_ in pair:  # warm-up: pools filled
                b_.run()
                b_.sync()
            nst = 6
            th = [None, None]
            self.barrier()
            t0 = time.perf_counter()
            for k in range(nst):
                j = k % 2
                if th[j] is not None:
                    th[j].join()  # this batch's slab has left: it may be overwritten
                pair[j].run()
                if k >= 1:
                    th[1 - j] = threading.Thread(target=gather, args=(1 - j,))
                    th[1 - j].start()
            last = (nst - 1) % 2
            if th[1 - last] is not None:
                th[1 - last].join()
            gather(last)
            self.barrier()
            overlapped = self.max((time.perf_counter() - t0) / nst * 1e3)
            if errs:
                raise RuntimeError(errs[0])
        c.close()
        return ms, overlapped

    def identities(self, J):
        """What makes an N-rank line self-verifying (VERDICT r5 "next" 7): for every rank the PCI bus id of the card
        it runs on (hipDeviceGetPCIBusId through jb_device_pci_bus_id) and the size RCCL itself reports for a
        communicator formed by the LIBRARY over all ranks (ncclCommCount through jb_comm_size) -- `n_gpus` of the line
        is WORLD_SIZE from the environment, which proves neither.  Two ranks on one card outside the one-GPU
        rehearsal (JB_BENCH_REHEARSE) end the run.  Returns (list of per-rank records, number of distinct cards)."""
        import ctypes

        buf = ctypes.create_string_buffer(64)
        bus = buf.value.decode() if J.lib().jb_device_pci_bus_id(self.local_rank, buf, 64) == 0 else None
        size, cerr, source = None, None, None
        lib_comm = bool(os.environ.get("JB_RCCL_LIBRARY")) or os.environ.get("JB_BENCH_VERIFY_COMM", "0") != "0"
        if self.dist is not None and not self.rehearse and not lib_comm:
            # the ranks that took part in an RCCL collective, counted BY the collective: a sum of ones over the process
            # group's communicator (backend nccl = RCCL).  The library's own communicator (ncclCommCount through
            # jb_comm_size) is asked with --gather, which forms one anyway, or with JB_BENCH_VERIFY_COMM=1: a second
            # communicator that failed to form would cost the driver's one multi-GPU run its line.
            one = self.torch.ones(1, dtype=self.torch.float64, device="cuda")
            self.dist.all_reduce(one)
            size, source = int(round(float(one.item()))), "all_reduce of ones over the process group's RCCL communicator"
        elif self.dist is not None and lib_comm:
            source = "ncclCommCount of a communicator formed by the library (jb_comm_size)"
            ids = [None]
            try:
                ids = [J.comm.unique_id() if self.rank == 0 else None]
            except Exception as e:  # noqa: BLE001  (every rank must still reach the broadcast)
                cerr = repr(e)
            self.dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                try:
                    c = J.comm.Comm(ids[0], self.world, self.rank, device=self.local_rank)
                    size = c.size()
                    c.close()
                except Exception as e:  # noqa: BLE001
                    cerr = repr(e)
        elif self.dist is None:
            size, source = 1, "one rank: no communicator"
        me = {"rank": self.rank, "local_rank": self.local_rank, "pci_bus_id": bus, "comm_size_from_rccl": size,
              "comm_size_source": source}
        if cerr:
            me["comm_error"] = cerr
        recs = [me]
        if self.dist is not None:
            recs = [None] * self.world
            self.dist.all_gather_object(recs, me)
        cards = {r["pci_bus_id"] for r in recs}
        if len(cards) != len(recs) and not self.rehearse:
            raise SystemExit(f"bench.py --gpus {self.world}: ranks share a GPU ({[r['pci_bus_id'] for r in recs]}); "
                             "one process per GPU is the contract (JB_BENCH_REHEARSE=1 for the one-GPU rehearsal)")
        return recs, len(cards)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


class Solo:
    """Rank 0 on its own, the other ranks idle at the next barrier: the N = 1 reference of a strong-scaling job,
    measured in the SAME run on the same box (run_config3 takes this in place of Ranks)."""

    def __init__(self, R):
        self.rank, self.world, self.local_rank, self.dry, self._t = 0, 1, R.local_rank, False, R.torch

    def barrier(self):
        self._t.cuda.synchronize()

    def max(self, x):
        return x

    def all_gather_floats(self, xs):
        return [list(xs)]


PROFILE_ROUND = "r06"  # profiles/<round>_* are what this line may quote


def kernel_sources_sha16():
    """Fingerprint of everything the kernels are built from (there is no .git on the GPU box): the
    profile files carry the fingerprint they were measured on, and a quote from a profile of other
    sources is withheld (`profiles_stale`)."""
    import hashlib

    h = hashlib.sha256()
    src = ROOT / "jbonsai_amd" / "csrc"
    for f in sorted(list(src.glob("*.hip")) + list(src.glob("*.h")) + list(src.glob("*.cpp")) + [src / "build.sh"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def read_profile(name, batch, frames):
    """(record, source, reason): a committed profile record of THIS workload on THESE sources, or None and
    why not.  Counters cannot be read inside an ordinary run; tools/traffic.sh and tools/pmc_voc.sh make the
    files from separate rocprofv3 --pmc passes of this same command."""
    f = ROOT / "profiles" / f"{PROFILE_ROUND}_{name}.json"
    if not f.exists():
        return None, None, f"profiles/{f.name} missing"
    try:
        j = json.loads(f.read_text())
    except Exception as e:
        return None, None, f"profiles/{f.name}: {e!r}"
    if j.get("batch") != batch or j.get("frames") != frames:
        return None, None, (f"profiles/{f.name} is for batch {j.get('batch')} x {j.get('frames')} frames, "
                            f"this run is {batch} x {frames}")
    if j.get("kernel_sources_sha16") != kernel_sources_sha16():
        return None, None, f"profiles/{f.name} was measured on other kernel sources (stale)"
    return j, f"profiles/{f.name}", None


def roofline_block(samples_per_launch, voc_ms, info, batch, frames):
    """The dominant kernel against the bound that binds it (FP64 VALU issue), with the HBM view the
    metric is defined on as the secondary record."""
    tflops = FLOP_PER_SAMPLE * samples_per_launch / (voc_ms * 1e-3) / 1e12
    gbs = B_ALG * samples_per_launch / (voc_ms * 1e-3) / 1e9
    stale = []
    sq, sq_src, why = read_profile("pmc_sq_k_vocoder_lt", batch, frames)
    sqrec = None
    if sq is not None and "SQ_ACTIVE_INST_VALU" in sq.get("counters", {}) and "SQ_WAVE_CYCLES" in sq["counters"]:
        c = sq["counters"]
        # two waves per SIMD share one VALU: busy = active / (wave cycles / 2)
        sqrec = {"source": f"{sq_src} (rocprofv3 --pmc, separate runs of this command on these sources)",
                 "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"], "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"],
                 "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"),
                 "valu_busy": c["SQ_ACTIVE_INST_VALU"] / (c["SQ_WAVE_CYCLES"] / 2.0)}
    elif why:
        sqrec = {"withheld": why}
        stale.append(why)
    tr, tr_src, why = read_profile("traffic", batch, frames)
    traffic = tr.get("hbm_bytes_per_launch") if tr is not None else None
    if tr is None and why:
        stale.append(why)
    return {
        "bound": "valu_f64", "kernel": info.get("kernel", "k_vocoder_lt"), "kernel_waves_per_simd": info.get("waves_per_simd"),
        "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
        "kernel_ms": voc_ms, "alg_flop_per_sample": FLOP_PER_SAMPLE, "path_flop_per_sample": PATH_FLOP_PER_SAMPLE,
        "note": "recursive IIR: bound by FP64 VALU issue, not by HBM (DESIGN.md section 4); achieved = useful f64 "
                "flops of what this kernel computes (V5-V9 of SURVEY 8a: 1.35 kflop per output sample; the whole "
                "path is 1.39) / kernel time.  Flop basis: rounds 1-4 priced this figure with 1.39 kflop (the path's), "
                "round 5 on with 1.35 (the kernel's): 3 % lower at the same kernel time",
        "sq_counters": sqrec,
        "traffic": traffic, "traffic_source": tr_src,
        "traffic_fetch_raw": tr.get("fetch_bytes_raw") if tr is not None else None,
        "whole_step_hbm_bytes": tr.get("whole_step_hbm_bytes") if tr is not None else None,
        "whole_step_fetch_bytes_raw": tr.get("whole_step_fetch_bytes_raw") if tr is not None else None,
        "traffic_note": ("FETCH_SIZE calibrated per access shape on known byte counts in the same rocprofv3 call "
                         "(tools/traffic.sh, tools/microbench/fetch_calib.hip); raw figures beside the corrected ones"
                         if tr is not None else None),
        "profiles_stale": bool(stale), "profiles_stale_why": stale or None,
        # the HBM view (SURVEY 8d's per-unit figure x samples per launch / kernel time against 8 TB/s)
        "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "alg_bytes_per_sample": B_ALG, "traffic": traffic},
    }


# ---------------------------------------------------------------------------------------------
def timed_steps(batches, steps, R):
    """EXACTLY `steps` steps over the resident batches, bracketed by barrier + synchronize; returns
    (seconds = max over ranks, vocoder kernel ms of every step)."""
    depth = len(batches)
    R.barrier()
    t0 = time.perf_counter()
    voc_ms = []
    inflight = [False] * depth
    for k in range(steps):
        # step k runs on batch k % depth; its stream is independent of the other batch's,
        # so parameter generation of this step overlaps the vocoder of the previous one
        j = k % depth
        if inflight[j]:
            batches[j].sync()
            voc_ms.append(batches[j].last_timing()[1])
        batches[j].run()
        inflight[j] = True
    for j in range(depth):
        if inflight[j]:
            batches[j].sync()
            voc_ms.append(batches[j].last_timing()[1])
    R.barrier()
    return R.max(time.perf_counter() - t0), voc_ms


def config3_shard(R, tab, n_utts):
    """This rank's share of the ONE config-3 job: the seed-fixed list of n_utts mixed lengths, LPT over
    the ranks, then LPT of the share into sub-batches that fit HBM.  Returns (lens of the whole list,
    [[IndexUtterance ...] per sub-batch], frames of this rank)."""
    from jbonsai_amd import shard, synth

    lens = synth.mixed_lengths(n_utts)  # the same list on every rank
    mine = shard.shard_for_rank(lens, R.rank, R.world)
    my_frames = sum(lens[i] for i in mine)
    k = max(1, -(-my_frames // SUB_BATCH_FRAMES))
    subs = shard.lpt_partition([lens[i] for i in mine], k)
    out = []
    for sb in subs:
        ids = [mine[j] for j in sb]
        out.append([synth.synth_utterance(tab, lens[i], 2000 + i, indexed=True) for i in ids])
    return lens, out, my_frames


def run_config3(R, J, tab, vi, pset, args, steps, warmup):
    """The strong-scaling job.  A step = the whole 4096-utterance list once: every rank walks its
    sub-batches; a sub-batch is created from pdf row indices (12 B per state cross PCIe; gather + blend
    on the device), run, certified and released INSIDE the timed region -- the shares of 1, 2 and 4
    ranks do not fit HBM as one batch.  The pdf tables are resident before it starts.
    A rank whose local work fails still joins every collective (the others must not hang on it) and
    the record carries the failure."""
    from jbonsai_amd import shard, synth

    lens = synth.mixed_lengths(args.utts)
    total_samples = sum(lens) * vi.fperiod
    err, subs, my_frames = None, [], 0
    try:
        lens, subs, my_frames = config3_shard(R, tab, args.utts)
    except Exception as e:
        err = repr(e)

    def run_passes(n):
        # n passes over this rank's sub-batches as one sequence, TWO in flight: while the GPU runs sub-batch i, the
        # next one (of this pass or the next) is created -- index upload, device gather, work list -- and its step
        # enqueued behind the running one, whose certification and redo round then run beside the next one's
        # parameter generation.  Two sub-batches resident at most, every creation inside the timed region.
        # (With the HIP runtime's default of four hardware queues this LOST -- 1027 against 910 ms per pass, the
        # second sub-batch's kernels queued in front of the first one's redo round; with the 16 the library asks
        # for now it wins: 834-855 against 895-938 ms per pass on one box.)
        seq = [k for _ in range(n) for k in range(len(subs))]

        def make(i):
            return J.Batch(vi, subs[seq[i]], device=R.local_rank, pdf_set=pset)

        if not seq:
            return
        cur = make(0)
        cur.run()
        for i in range(len(seq)):
            nb = None
            try:
                if i + 1 < len(seq):
                    nb = make(i + 1)
                    nb.run()
                cur.sync()
            except Exception:
                if nb is not None:
                    nb.close()
                raise
            finally:
                cur.close()
            cur = nb

    def passes(n):
        nonlocal err
        if err is not None or R.dry:
            return
        try:
            run_passes(n)
        except Exception as e:
            err = repr(e)

    passes(max(warmup, 0))
    R.barrier()
    t0 = time.perf_counter()
    passes(steps)
    t_mine = time.perf_counter() - t0
    R.barrier()
    dt = R.max(time.perf_counter() - t0)
    per_rank = R.all_gather_floats([t_mine / steps * 1e3, float(my_frames), float(len(subs)),
                                    0.0 if err is None else 1.0])
    parts = shard.lpt_partition(lens, R.world)
    rec = {
        "workload": f"ONE job of {args.utts} distinct synthetic utterances, lengths U[400, 25546] frames (seed-fixed, "
                    f"BASELINE config 3), LPT-sharded over {R.world} rank(s); each rank walks its share in "
                    f"sub-batches of <= {SUB_BATCH_FRAMES} frames created from pdf row indices, run and released "
                    "inside the timed region",
        "value": total_samples * steps / dt, "unit": "samples/s", "scaling": "strong",
        "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
        "realtime_factor": total_samples * steps / dt / vi.sampling_frequency,
        "total_samples_per_step": total_samples,
        "per_rank_ms": [p[0] for p in per_rank], "per_rank_frames": [int(p[1]) for p in per_rank],
        "per_rank_sub_batches": [int(p[2]) for p in per_rank],
        "imbalance_max_over_mean_frames": shard.imbalance(lens, parts),
    }
    if any(p[3] != 0.0 for p in per_rank):
        rec["error"] = f"rank(s) {[i for i, p in enumerate(per_rank) if p[3] != 0.0]} failed" + (f": {err}" if err else "")
        rec["value"] = None
    return rec


def extras_single_gpu(J, eng, tab, vi, args, batch, batch_utts, frames, ms_per_step, R):
    """Secondary records of the N = 1 line; none of them enters `value`."""
    import numpy as np
    from jbonsai_amd import synth
    from tests.golden.labels import SAMPLE_SENTENCE_2

    ex = {}

    def host_mem_ok(nbytes):
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) * 1024 > 3 * nbytes
        except OSError:
            pass
        return False

    # (0) two batches in flight (--pipeline 2): one batch's parameter generation beside the other's vocoder.
    # FIRST of the extras and on two fresh batches: behind the D2H extra below, or with the batch that went
    # through it, the figure came out 4-8 ms above what `--pipeline 2` gives on the same box
    try:
        pair = [J.Batch(vi, batch_utts, device=R.local_rank) for _ in range(2)]
        for _ in range(2):
            for b_ in pair:
                b_.run()
            for b_ in pair:
                b_.sync()
        t0 = time.perf_counter()
        nst = 8
        for k in range(nst):
            if k >= 2:
                pair[k % 2].sync()
            pair[k % 2].run()
        for b_ in pair:
            b_.sync()
        ex["two_batches_in_flight"] = {"ms_per_step": (time.perf_counter() - t0) / nst * 1e3}
        for b_ in pair:
            b_.close()
    except Exception as e:
        ex["two_batches_in_flight"] = {"error": repr(e)}
    # (1) PCM on the host: the step plus the staged D2H of the whole slab (f64, and the fused 16-bit sink)
    try:
        if not host_mem_ok(batch.total_samples * 8):
            raise MemoryError("host memory too small for the PCM of this batch")
        bufs = [np.empty(batch.num_samples(i), dtype=np.float64) for i in range(len(batch))]
        batch.pcm_all(bufs)  # first read touches the pages; the timed one below is at link rate
        t0 = time.perf_counter()
        batch.pcm_all(bufs)
        d2h = (time.perf_counter() - t0) * 1e3
        nbytes = sum(b.nbytes for b in bufs)
        ex["d2h_inclusive"] = {"f64": {"d2h_ms": d2h, "ms_per_step": ms_per_step + d2h, "GBps": nbytes / d2h / 1e6,
                                       "bytes": nbytes}}
        del bufs
        b16 = J.Batch(vi, batch_utts, device=R.local_rank, pcm_i16=True)
        b16.run()
        b16.sync()
        t0 = time.perf_counter()
        for _ in range(2):
            b16.run()
            b16.sync()
        step16 = (time.perf_counter() - t0) / 2 * 1e3
        bufs = [np.empty(b16.num_samples(i), dtype=np.int16) for i in range(len(b16))]
        b16.pcm_all(bufs)
        t0 = time.perf_counter()
        b16.pcm_all(bufs)
        d2h = (time.perf_counter() - t0) * 1e3
        ex["d2h_inclusive"]["i16"] = {"d2h_ms": d2h, "ms_per_step": step16 + d2h, "step_ms": step16,
                                      "GBps": sum(b.nbytes for b in bufs) / d2h / 1e6}
        del bufs
        b16.close()
    except Exception as e:  # a secondary record must not cost the headline
        ex["d2h_inclusive"] = {"error": repr(e)}
    # (1b) host-visible throughput (Engine::synthesize hands PCM to the host, src/engine.rs:294): two batches
    #      alternating, the staged D2H of step k's PCM (its own copy stream, a reader thread) running beside the GPU
    #      work of step k + 1.  16-bit sink: the 3.1 GB copy is shorter than a step and hides behind it; f64: 12.6 GB
    #      over the link take longer than a step, so the link rate is the bound -- both stated.
    def host_visible(dtype, i16):
        import threading

        pair = [J.Batch(vi, batch_utts, device=R.local_rank, pcm_i16=i16) for _ in range(2)]
        try:
            bufs = [[np.empty(b_.num_samples(i), dtype=dtype) for i in range(len(b_))] for b_ in pair]
            for b_, bf in zip(pair, bufs):  # warm-up: pages touched, pools filled
                b_.run()
                b_.pcm_all(bf)
            errs = []

            def read(j):
                try:
                    pair[j].pcm_all(bufs[j])  # waits for that batch's step and its certification, then copies
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))

            nst = 6
            readers = [None, None]
            t0 = time.perf_counter()
            for k in range(nst):
                j = k % 2
                if readers[j] is not None:
                    readers[j].join()  # the slab of this batch has been read: it may be overwritten
                pair[j].run()
                if k >= 1:
                    readers[1 - j] = threading.Thread(target=read, args=(1 - j,))
                    readers[1 - j].start()
            last = (nst - 1) % 2
            if readers[1 - last] is not None:
                readers[1 - last].join()
            read(last)
            dt = (time.perf_counter() - t0) / nst * 1e3
            if errs:
                raise RuntimeError(errs[0])
            nbytes = sum(b.nbytes for b in bufs[0])
            return {"ms_per_step": dt, "value": batch.total_samples / (dt * 1e-3), "unit": "samples/s on the host",
                    "bytes_per_step": nbytes, "link_GBps_if_link_bound": nbytes / dt / 1e6, "steps": nst,
                    "how": "two batches alternate; D2H of step k on a copy stream beside the GPU work of step k+1"}
        finally:
            for b_ in pair:
                b_.close()

    try:
        ex["host_visible"] = {"i16": host_visible(np.int16, True)}
        if host_mem_ok(3 * batch.total_samples * 8):
            ex["host_visible"]["f64"] = host_visible(np.float64, False)
            ex["host_visible"]["f64"]["note"] = "12.6 GB per step over PCIe: link-bound (the copy is longer than a step)"
        else:
            ex["host_visible"]["f64"] = {"skipped": "host memory too small for two f64 PCM sets"}
    except Exception as e:
        ex.setdefault("host_visible", {})["error"] = repr(e)
    # (1c) the headline's family: config 2 as copies of the utterance of seeds 0-3.  Copies of ONE utterance have the
    #      same hand-off positions in every copy: a position that fails its certification fails 256 times and costs a
    #      redo round, one that passes costs nothing -- the line above is ONE member of this family (seed 0).
    try:
        fam = []
        for seed in range(4):
            u = batch_utts[0] if seed == args.seed else synth.synth_utterance(tab, frames, seed)
            with J.Batch(vi, [u] * args.batch, device=R.local_rank) as bs:
                for _ in range(2):
                    bs.run()
                    bs.sync()
                # (the median of individually timed steps: the first steps behind the host-visible leg's 25 GB of pinned
                #  buffers have come out 5 ms long on some boxes, and a mean of six carries that into the family's mean)
                ts = []
                for _ in range(6):
                    t0 = time.perf_counter()
                    bs.run()
                    bs.sync()
                    ts.append((time.perf_counter() - t0) * 1e3)
                ms = float(np.median(ts))
                inf, rs = bs.info(), bs.redo_stats()
                fam.append({"seed": seed, "ms_per_step": ms, "max_ms": max(ts), "chunks_redone": inf["n_redo"],
                            "settled_at_checkpoint": rs[0], "redone_to_end": rs[1],
                            "warmup_frames": inf["warmup_frames"], "chunk_frames": inf["chunk_frames"]})
        ex["copies_over_seeds"] = {"per_seed": fam, "steps": 6, "statistic": "median of 6 individually timed steps",
                                   "mean_ms_per_step": sum(f["ms_per_step"] for f in fam) / len(fam),
                                   "mean_value": batch.total_samples / (sum(f["ms_per_step"] for f in fam) / len(fam) * 1e-3),
                                   "unit": "samples/s",
                                   "note": "BASELINE config 2 for four different 25,546-frame utterances (256 copies each); "
                                           "`value` of this line is seed 0"}
    except Exception as e:
        ex["copies_over_seeds"] = {"error": repr(e)}
    # (2) config 2 with 64 DISTINCT utterances tiled over the batch: real hand-off failures and redo
    try:
        nd = 64
        du = [batch_utts[0]] + [synth.synth_utterance(tab, frames, 1000 + i) for i in range(1, nd)]
        bd = J.Batch(vi, [du[i % nd] for i in range(args.batch)], device=R.local_rank)
        for _ in range(2):
            bd.run()
            bd.sync()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            bd.run()
            bd.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        ex["distinct_64"] = {"ms_per_step": float(np.median(ts)), "max_ms": max(ts), "steps": 8,
                             "statistic": "median of 8 individually timed steps",
                             "chunks_redone": bd.info()["n_redo"],
                             "settled_at_checkpoint": bd.redo_stats()[0], "redone_to_end": bd.redo_stats()[1]}
        bd.close()
    except Exception as e:
        ex["distinct_64"] = {"error": repr(e)}
    # (3) labels -> PCM on the host through the engine entry (front half on host threads, device gather,
    #     GPU hot path, staged D2H): 64 utterances of 75 x SAMPLE_SENTENCE_2
    try:
        utt = list(SAMPLE_SENTENCE_2) * 75
        lb = [utt] * 64
        eng.synthesize_batch(lb[:2])
        rec = {}
        for key, i16 in (("f64", False), ("i16", True)):
            # two calls: the first one of a shape also pays for device and pinned-host allocations that the
            # library's pools keep afterwards (it took 208-690 ms depending on what ran before it)
            walls = []
            for _ in range(2):
                outs = None
                t0 = time.perf_counter()
                outs = eng.synthesize_batch(lb, i16=i16)
                walls.append(time.perf_counter() - t0)
            dt = walls[1]
            ns = sum(len(o) for o in outs)
            rec[key] = {"wall_ms": dt * 1e3, "first_call_ms": walls[0] * 1e3, "samples": ns,
                        "realtime_factor": ns / dt / vi.sampling_frequency}
        rec["workload"] = f"jb_synthesize_batch: 64 utterances x {len(outs[0]) / vi.sampling_frequency:.0f} s (labels in, PCM on the host out)"
        outs = None
        # the reference tree's only long label sequence: examples/genji/genji.lab, 1,456 distinct labels
        # (164 s with the nitech voice), 64 copies of it as one request
        from tests.golden.labels import GENJI

        gl = [list(GENJI)] * 64
        eng.synthesize_batch(gl[:2])
        walls = []
        for _ in range(2):
            outs = None
            t0 = time.perf_counter()
            outs = eng.synthesize_batch(gl)
            walls.append(time.perf_counter() - t0)
        ns = sum(len(o) for o in outs)
        rec["genji_1456_labels"] = {"workload": f"jb_synthesize_batch: 64 x genji.lab (1,456 labels, "
                                                f"{len(outs[0]) / vi.sampling_frequency:.0f} s each), f64 PCM on the host",
                                    "wall_ms": walls[1] * 1e3, "first_call_ms": walls[0] * 1e3, "samples": ns,
                                    "realtime_factor": ns / walls[1] / vi.sampling_frequency}
        outs = None
        # the reference's OWN benchmark (benches/bonsais.rs:11-140): Engine::synthesize of one sentence -- 8, 20 and 43
        # labels -- labels in, f64 PCM on the host out; warm, median of 20 calls.  The oracle's time for the same
        # call is added beside it by the cpu_baseline leg ("oracle_ms").
        from tests.golden.labels import BENCH_LETTER, SAMPLE_SENTENCE_1

        single = {}
        for name, lab in (("bonsai_8_labels", SAMPLE_SENTENCE_1), ("is_bonsai_20_labels", SAMPLE_SENTENCE_2),
                          ("bonsai_letter_43_labels", BENCH_LETTER)):
            eng.synthesize(lab)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                o = eng.synthesize(lab)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            single[name] = {"labels": len(lab), "samples": int(len(o)), "audio_s": len(o) / vi.sampling_frequency,
                            "median_ms": ts[len(ts) // 2], "min_ms": ts[0], "calls": 20,
                            "realtime_factor": len(o) / vi.sampling_frequency / (ts[len(ts) // 2] * 1e-3)}
        rec["single"] = single
        rec["single_note"] = ("jb_synthesize, one sentence per call (the three of the reference's benches/bonsais.rs), warm, "
                              "f64 PCM on the host; a latency, not a throughput: one sentence cannot fill the chip")
        ex["labels_to_pcm"] = rec
    except Exception as e:
        ex["labels_to_pcm"] = {"error": repr(e)}
    # (4) BASELINE configs 4 and 5 at their stated batch of 1024, resident like the headline: DISTINCT
    #     synthetic utterances of the same total length as config 2 (1024 x 6,386 frames = 31.9 s each),
    #     created from pdf row indices.  Config 4's voice (tohoku-f01) is not in the reference tree: the
    #     documented substitute is nitech with its three streams.  Config 5 blends TWO DIFFERENT
    #     voices on the device (nitech + the permuted nitech of tests/golden/make_permuted_voice.py, 0.5/0.5).
    try:
        ex["config4"] = resident_record(J, vi, R, 1024, [tab], None, 4000,
                                        "BASELINE config 4 (nitech-LPF substitute for the absent tohoku-f01 voice): "
                                        "batch=1024 distinct synthetic utterances x 6386 frames, MCP+LF0+LPF streams")
    except Exception as e:
        ex["config4"] = {"error": repr(e)}
    try:
        import tempfile

        from tests.golden.make_permuted_voice import permuted_voice_path

        with tempfile.TemporaryDirectory() as td:
            eng2 = J.Engine.load([VOICE, permuted_voice_path(td)])
        tabs = [synth.VoiceTables(eng2, 0), synth.VoiceTables(eng2, 1)]
        half = {"duration": [0.5, 0.5], "parameter": [[0.5, 0.5]] * 3, "gv": [[0.5, 0.5]] * 3}
        ex["config5"] = resident_record(J, eng2.voice_info(), R, 1024, tabs, half, 5000,
                                        "BASELINE config 5: two-voice interpolation 0.5/0.5 (nitech + permuted nitech: "
                                        "the tohoku-f01 files are absent), batch=1024 distinct synthetic utterances x "
                                        "6386 frames, rows of both voices gathered and blended on the device")
        eng2.close()
    except Exception as e:
        ex["config5"] = {"error": repr(e)}
    # (5) another mel-cepstral order at the same shape (round 6): the reference is generic in the order
    #     (vocoder/mod.rs:45-70, mlsa.rs:38-45), the throughput kernel was built for nitech's two until round 6.  An
    #     order-49 voice (vector length 50: nitech's MCP stream widened by seeded small dimensions, synth.with_order),
    #     1024 x 6,386 frames as 64 distinct utterances x 16 copies, state-level (the widened Gaussians are not rows
    #     of a voice file), resident like the headline.  The order-34 figure beside it is config 4's.
    try:
        ex["order49"] = order_record(J, vi, R, tab, 50, 1024, 64)
        c4 = ex.get("config4", {}).get("ms_per_step")
        if c4:
            ex["order49"]["vs_config4_scaled_by_order"] = ex["order49"]["ms_per_step"] / (c4 * 49.0 / 34.0)
    except Exception as e:
        ex["order49"] = {"error": repr(e)}
    # (6) what the reference's bitwise-stability claim costs here (README.md:65,124; VERDICT r5 "next" 5): config 2 with
    #     jb_engine_set_batch_invariant's flags (JB_BATCH_SERIAL | JB_BATCH_SERIAL_GV) -- every utterance one unchunked
    #     recursion on one wave, GV sums in the reference's serial order: the same bits alone, in any batch, on any
    #     device count.  One warm-up step and two timed ones (a step is seconds, not milliseconds).
    try:
        with J.Batch(vi, batch_utts, device=R.local_rank, serial=True, serial_gv=True) as bi:
            bi.run()
            bi.sync()
            t0 = time.perf_counter()
            for _ in range(2):
                bi.run()
                bi.sync()
            ms = (time.perf_counter() - t0) / 2 * 1e3
            ex["batch_invariant"] = {"ms_per_step": ms, "value": bi.total_samples / (ms * 1e-3), "unit": "samples/s",
                                     "realtime_factor": bi.total_samples / (ms * 1e-3) / vi.sampling_frequency,
                                     "vs_headline": ms / ms_per_step, "steps": 2,
                                     "how": "JB_BATCH_SERIAL | JB_BATCH_SERIAL_GV on the headline's batch: one wave per "
                                            "utterance, no time-chunks; bits independent of the batch"}
    except Exception as e:
        ex["batch_invariant"] = {"error": repr(e)}
    return ex


CONFIG45_FRAMES = 6386  # 1024 x 6386 = 6,539,264 frames: config 2's total (256 x 25,546 = 6,539,776)


def resident_record(J, vi, R, n_utts, tabs, weights, id0, workload, steps=6):
    """One resident batch of n_utts distinct synthetic utterances created from pdf row indices (one voice:
    tabs = [tab]; several: rows of every voice + weights), timed like the headline: `steps` steps after
    two warm-up steps, inputs resident.  Carries its own ms_per_step, redo counts and roofline fraction."""
    from jbonsai_amd import synth

    pset = synth.voice_set_pdf_set(tabs, R.local_rank)
    try:
        if weights is None:
            utts = [synth.synth_utterance(tabs[0], CONFIG45_FRAMES, id0 + i, indexed=True) for i in range(n_utts)]
        else:
            utts = [synth.synth_utterance_voices(tabs, weights, CONFIG45_FRAMES, id0 + i, indexed=True)
                    for i in range(n_utts)]
        t0 = time.perf_counter()
        b = J.Batch(vi, utts, device=R.local_rank, pdf_set=pset)
        create_ms = (time.perf_counter() - t0) * 1e3
        try:
            for _ in range(2):
                b.run()
                b.sync()
            voc = []
            t0 = time.perf_counter()
            for _ in range(steps):
                b.run()
                b.sync()
                voc.append(b.last_timing()[1])
            ms = (time.perf_counter() - t0) / steps * 1e3
            info, redo = b.info(), b.redo_stats()
            ns = b.total_samples
        finally:
            b.close()
    finally:
        pset.close()
    voc_ms = sum(voc) / len(voc)
    tflops = FLOP_PER_SAMPLE * ns / (voc_ms * 1e-3) / 1e12
    return {"workload": workload, "batch": n_utts, "frames_per_utterance": CONFIG45_FRAMES, "voices": len(tabs),
            "ms_per_step": ms, "value": ns / (ms * 1e-3), "unit": "samples/s", "steps": steps,
            "realtime_factor": ns / (ms * 1e-3) / vi.sampling_frequency, "samples_per_step": ns,
            "create_ms_from_row_indices": create_ms,
            "vocoder_chunk_frames": info["chunk_frames"], "vocoder_work_items": info["n_items"],
            "chunks_redone_last_step": info["n_redo"], "settled_at_checkpoint": redo[0], "redone_to_end": redo[1],
            "roofline": {"bound": "valu_f64", "kernel_ms": voc_ms, "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS}}


def order_record(J, vi, R, tab, nmcp, n_utts, distinct, steps=6):
    """One resident batch of `n_utts` state-level utterances of ANOTHER mel-cepstral order (vector length nmcp),
    `distinct` different ones tiled over the batch, timed like the headline."""
    from jbonsai_amd import synth

    pairs = [synth.with_order(vi, synth.synth_utterance(tab, CONFIG45_FRAMES, 6000 + i), nmcp, seed=300 + i)
             for i in range(distinct)]
    vi2 = pairs[0][0]
    with J.Batch(vi2, [pairs[i % distinct][1] for i in range(n_utts)], device=R.local_rank) as b:
        for _ in range(2):
            b.run()
            b.sync()
        voc = []
        t0 = time.perf_counter()
        for _ in range(steps):
            b.run()
            b.sync()
            voc.append(b.last_timing()[1])
        ms = (time.perf_counter() - t0) / steps * 1e3
        info, redo, ns = b.info(), b.redo_stats(), b.total_samples
        kern, waves = b.kernel_info()
    voc_ms = sum(voc) / len(voc)
    # flops of V5-V9 scale with the taps: 1.35 kflop at 34 taps -> x (nmcp - 1) / 34
    tflops = FLOP_PER_SAMPLE * (nmcp - 1) / 34.0 * ns / (voc_ms * 1e-3) / 1e12
    return {"workload": f"order-{nmcp - 1} voice (nitech's MCP stream widened to {nmcp} dims): batch={n_utts} x "
                        f"{CONFIG45_FRAMES} frames, {distinct} distinct utterances tiled, state-level, resident",
            "batch": n_utts, "frames_per_utterance": CONFIG45_FRAMES, "nmcp": nmcp, "kernel": kern,
            "waves_per_simd": waves, "ms_per_step": ms, "value": ns / (ms * 1e-3), "unit": "samples/s", "steps": steps,
            "realtime_factor": ns / (ms * 1e-3) / vi.sampling_frequency, "samples_per_step": ns,
            "vocoder_chunk_frames": info["chunk_frames"], "vocoder_work_items": info["n_items"],
            "chunks_redone_last_step": info["n_redo"], "settled_at_checkpoint": redo[0], "redone_to_end": redo[1],
            "roofline": {"bound": "valu_f64", "kernel_ms": voc_ms, "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                         "flop_per_sample": FLOP_PER_SAMPLE * (nmcp - 1) / 34.0}}


def config3_n1_reference(R, J, tab, vi, pset, args):
    """At N > 1: the SAME 4096-utterance job on rank 0 alone (one warm-up pass, two timed), the other ranks idle --
    so that the strong-scaling figure of a line is one division of two numbers of the same run and box."""
    if R.world == 1 or R.dry:
        return None
    rec = None
    if R.rank == 0:
        try:
            rec = run_config3(Solo(R), J, tab, vi, pset, args, 2, 1)
            rec = {k: rec.get(k) for k in ("value", "unit", "ms_per_step", "steps", "error") if k in rec}
            rec["how"] = "rank 0 alone right after the N-rank passes, the other ranks idle at a barrier"
        except Exception as e:  # noqa: BLE001
            rec = {"error": repr(e)}
    R.barrier()
    return rec


def run_rank(args):
    # the config-3 job keeps two sub-batches alive (one running, the next being created): let the
    # library's device-memory pool hold both sets of blocks between passes (default cap 64 GB).  Not for
    # the single-GPU default run: its secondary measurements create differently shaped batches, and a
    # pool that fills the device turns their allocations into out-of-memory retries
    if args.job == "config3":
        os.environ.setdefault("JB_DEVICE_POOL_MB", "160000")
    import torch  # first: so that this process uses ONE HIP runtime (same SONAME as ours)

    R = Ranks()
    import jbonsai_amd as J
    from jbonsai_amd import synth

    eng = J.Engine.load([VOICE])
    tab = synth.VoiceTables(eng)
    vi = eng.voice_info()
    vi.beta = args.beta
    frames = args.frames or synth.T_128S
    out = None

    if R.dry:
        time.sleep(float(os.environ.get("JB_BENCH_DRYRUN_SLEEP_S", "0")))  # launcher tests: a rank that hangs
        rec = run_config3(R, J, tab, vi, None, args, 1, 0)
        gms = R.gather_slabs(torch.arange(1000 * (R.rank + 1), dtype=torch.float64))
        ranks, n_cards = R.identities(J)  # (no GPU here: bus ids are None, no communicator is formed -- the plumbing)
        if R.rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": R.world, "gather_ms": gms, "config3_plan": rec, "ranks": ranks,
                              "distinct_gpus": n_cards}), flush=True)
        R.close()
        return
    if args.job == "config3":
        pset = tab.pdf_set(R.local_rank)
        ranks, n_cards = R.identities(J)
        rec = run_config3(R, J, tab, vi, pset, args, args.steps, args.warmup)
        n1 = config3_n1_reference(R, J, tab, vi, pset, args)
        if R.rank == 0:
            out = {
                "metric": "48 kHz PCM samples/sec (whole node), batched utterances",
                "value": rec["value"], "unit": "samples/s", "n_gpus": R.world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": rec["workload"], "utterances": args.utts,
                           "parallelism": f"utterance-sharded x{R.world} (LPT by frames)"},
                "realtime_factor": rec["realtime_factor"],
                "per_rank_ms": rec["per_rank_ms"], "per_rank_frames": rec["per_rank_frames"],
                "per_rank_sub_batches": rec["per_rank_sub_batches"],
                "imbalance_max_over_mean_frames": rec["imbalance_max_over_mean_frames"],
                "ranks": ranks, "distinct_gpus": n_cards,
            }
            if n1 is not None:
                out["config3_job_n1"] = n1
                if n1.get("value") and rec.get("value"):
                    out["strong_scaling_speedup"] = rec["value"] / n1["value"]
            print(json.dumps(out), flush=True)
        pset.close()
        R.close()
        return

    # ---- config 2 (headline): every utterance of the batch is the same sequence ("256 copies"),
    # uploaded once and aliased; outputs / workspace / filter state are per utterance
    utt = synth.synth_utterance(tab, frames, args.seed)
    nd = max(1, min(args.distinct, args.batch))
    utts = [utt] + [synth.synth_utterance(tab, frames, 1000 + i) for i in range(1, nd)]
    depth = max(1, args.pipeline)
    if args.mixed:
        lens = synth.mixed_lengths(args.batch, seed=3 + R.rank)
        batch_utts = [synth.synth_utterance(tab, T, 2000 + i) for i, T in enumerate(lens)]
    else:
        batch_utts = [utts[i % nd] for i in range(args.batch)]
    batches = [J.Batch(vi, batch_utts, device=R.local_rank, kernel=args.kernel, chunk_frames=args.chunk_frames,
                       warmup_frames=args.warmup_frames) for _ in range(depth)]
    batch = batches[0]
    samples_per_step = batch.total_samples

    for _ in range(max(args.warmup, 0)):
        for b_ in batches:
            b_.run()
        for b_ in batches:
            b_.sync()
    dt, voc_ms = timed_steps(batches, args.steps, R)

    gather_ms = gather_ovl = None
    gather_dtype = "f64" if args.gather_f64 else "i16"
    if args.gather and R.world > 1:
        # optional sink of north_star: PCM of all ranks on GPU 0 (rank 0 needs world x 12.6 GB of HBM for
        # config 2).  The library's own gather: grouped ncclSend / ncclRecv over xGMI (jb_gather_pcm); the
        # communicator id is control plane and travels over the launcher's rendezvous.  In the one-GPU
        # rehearsal (gloo) RCCL cannot run: there the slabs go through torch.distributed on the host.
        # With JB_RCCL_LIBRARY = the test double of tests/fake_rccl (several ranks on one device) the rehearsal
        # takes the library's own gather too: functional only, its time says nothing about xGMI.
        # Default: the 16-bit slab (north_star's sink is PCM; 3.1 GB per rank instead of 12.6), also measured beside
        # the next step: two batches alternate, as in `host_visible`.
        if R.rehearse and not os.environ.get("JB_RCCL_LIBRARY"):
            gather_ms = R.gather_slabs(pcm_slab_tensor(batch))
            gather_dtype = "f64"
        else:
            i16 = not args.gather_f64
            pair = [J.Batch(vi, batch_utts, device=R.local_rank, pcm_i16=i16) for _ in range(2)]
            try:
                pair[0].run()
                pair[0].sync()
                gather_ms, gather_ovl = R.gather_native(J, pair[0], pair)
            finally:
                for b_ in pair:
                    b_.close()

    ranks, n_cards = R.identities(J)
    info = batch.info()
    info["kernel"], info["waves_per_simd"] = batch.kernel_info()
    redo_stats = batch.redo_stats()
    ms_per_step = dt / args.steps * 1e3
    for b_ in batches[1:]:
        b_.close()

    if R.rank == 0:
        total = samples_per_step * R.world * args.steps
        value = total / dt
        voc_avg_ms = sum(voc_ms) / len(voc_ms)
        out = {
            "metric": "48 kHz PCM samples/sec (whole node), batched utterances",
            "value": value, "unit": "samples/s", "n_gpus": R.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"batch={args.batch}/GPU distinct synthetic state-level utterances of mixed length "
                             "U[400, 25546] frames from real nitech pdfs (BASELINE config 3 share), nitech voice"
                             if args.mixed else
                             f"batch={args.batch}/GPU copies of a {frames}-frame "
                             f"({frames * vi.fperiod / vi.sampling_frequency:.1f} s) synthetic state-level "
                             "utterance from real nitech pdfs (BASELINE config 2), nitech voice"),
                "batch_per_gpu": args.batch, "frames_per_utterance": frames,
                "samples_per_step_per_gpu": samples_per_step, "parallelism": f"utterance-sharded x{R.world}",
                "batches_in_flight": depth, "distinct_utterances": nd, "utterance_seed": args.seed,
                "beta": args.beta,
                "chunks_settled_at_checkpoint_last_step": redo_stats[0],
                "vocoder_chunk_frames": info["chunk_frames"], "vocoder_warmup_frames": info["warmup_frames"],
                "vocoder_work_items": info["n_items"], "chunks_redone_last_step": info["n_redo"],
            },
            "realtime_factor": value / vi.sampling_frequency,
            "ranks": ranks, "distinct_gpus": n_cards,
            # (the key carries the slab's type since round 6: rounds 1-4 reported the f64 slab of the timed batch as
            #  `gather_ms`, round 5 the 16-bit slab of a fresh pair under the same key -- not comparable: ADVICE r5)
            **({f"gather_{gather_dtype}_ms": gather_ms, "gather_overlapped_ms_per_step": gather_ovl, "gather_dtype": gather_dtype,
                "gather_bytes_into_root": (R.world - 1) * samples_per_step * (8 if gather_dtype == "f64" else 2),
                "gather_comm_size_from_rccl": getattr(R, "gather_comm_size", None)}
               if gather_ms is not None else {}),
            "roofline": roofline_block(samples_per_step, voc_avg_ms, info, args.batch, frames),
            "kernel_sources_sha16": kernel_sources_sha16(),
        }
    gpu_out = None
    if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline and not args.mixed and nd == 1:
        # what the error figure compares: PCM of the first and the last utterance of the timed batch, and the
        # excitation of the same utterance from a one-utterance batch with the debug tap
        gpu_out = {"pcm": [batch.pcm(0), batch.pcm(args.batch - 1)], "exc": None}
        try:
            with J.Batch(vi, [utt], device=R.local_rank, keep_tracks=True) as bt:
                bt.run()
                bt.sync()
                gpu_out["exc"] = bt.excitation(0)
        except Exception as e:  # the figure then lacks the excitation part
            print(f"excitation tap failed: {e!r}", file=sys.stderr)
    if R.world == 1 and not args.no_extras:
        ex = extras_single_gpu(J, eng, tab, vi, args, batch, batch_utts, frames, ms_per_step, R)
        if out is not None:
            out.update(ex)
    batch.close()
    if not args.no_extras:
        # BASELINE config 3 as ONE job beside the headline: the 4096-utterance list, this rank's share walked
        # in sub-batches: one warm-up pass and two timed passes ("config3_strong" at N > 1, "config3_job" at
        # N = 1: ~0.9 s per pass).  The job keeps two sub-batches alive: let the memory pool hold both
        pset = None
        try:
            J.lib().jb_release_cached_memory()
            J.lib().jb_set_cached_memory_limit(160000)
            pset = tab.pdf_set(R.local_rank)
        except Exception:
            pass  # run_config3 reports it: creating a batch over a missing set fails on this rank only
        rec = run_config3(R, J, tab, vi, pset, args, 2, 1)
        n1 = config3_n1_reference(R, J, tab, vi, pset, args) if pset is not None else None
        if pset is not None:
            pset.close()
        if out is not None:
            out["config3_strong" if R.world > 1 else "config3_job"] = rec
            if n1 is not None:
                # the N = 1 value of the same job beside the N-rank one: strong scaling is one division
                out["config3_job_n1"] = n1
                if n1.get("value") and rec.get("value"):
                    out["config3_strong"]["speedup_over_n1"] = rec["value"] / n1["value"]
    if R.rank == 0:
        if R.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(utt, vi, args.batch, gpu_out)
            # the metric's error figure, at the top level of the line (it is part of the metric)
            out["error_vs_oracle"] = out["cpu_baseline"].pop("error_vs_oracle")
            # the oracle's time for the reference's own benchmark sentences, beside the GPU engine's
            sm = out["cpu_baseline"].get("single_sentence_ms") or {}
            for name, rec_ in (out.get("labels_to_pcm", {}).get("single") or {}).items():
                if isinstance(sm.get(name), float):
                    rec_["oracle_ms"] = sm[name]
        print(json.dumps(out), flush=True)
    R.close()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no torchrun around us: be the launcher.  Nothing above has imported torch or touched HIP.
        sys.exit(spawn_ranks(args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
