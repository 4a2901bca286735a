#!/usr/bin/env python3
"""bench.py -- 48 kHz PCM samples/s of the parameter-generation + MLSA-vocoder hot
path on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one pass of the whole hot path (MLPG+GV x3 -> frame prologue -> pulse
schedule -> excitation + MLSA cascade) over one batch of synthetic utterances
whose state-level inputs are already resident in HBM.  N=1 workload = BASELINE
config 2: batch of 256 copies of the ~128 s utterance (T = 25,546 frames,
6,131,040 samples each) built from real nitech pdfs (jbonsai_amd/synth.py).
N>1: every rank runs the same per-GPU workload (weak scaling, no data-path
collective: utterances are independent); RCCL is used only for the barrier and
the max-over-ranks of the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

B_ALG = 8.67          # algorithmic bytes per output sample, f64 PCM
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FLOP_PER_SAMPLE = 1.39e3      # algorithmic f64 flops per output sample
# FP64 vector peak = half the guide's 157.3 TFLOPS FP32 vector rate (1024 SIMDs x 16 lanes x 2 flop x
# 2.4 GHz); tools/microbench/f64_rate.hip sustains 68 TFLOP/s of dependent-free v_fma_f64 on this part
FP64_VALU_PEAK_TFLOPS = 78.6
DEFAULT_CU_SPLIT = 0  # CU partition off: measured slower at every split
VOICE = ROOT / "tests" / "golden" / "voice" / "nitech_jp_atr503_m001.htsvoice"


def cpu_baseline(utt, vi, n_utts_per_thread=2):
    """The oracle (C restatement of jbonsai's CPU path, kind="port") on the host cores
    of this box, on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("JB_CPU_THREADS", "16"))))  # one GPU's CPU share
    sts = []
    for i, s in enumerate(utt.streams):
        si = vi.streams[i]
        msd = s.msd if s.msd is not None else np.full(len(utt.durations), 1.7976931348623157e308)
        sts.append(O.StreamStates(si.vector_length, len(si.windows), si.is_msd, si.use_gv,
                                  [len(w) for w in si.windows], [c for w in si.windows for c in w],
                                  s.mean, s.var, msd, s.gv_mean, s.gv_var, s.gv_switch))
    O.lib()
    nsamp = int(utt.durations.sum()) * vi.fperiod

    def work():
        for _ in range(n_utts_per_thread):
            tr = [O.mlpg(s, utt.durations) for s in sts]
            O.vocoder(vi.sampling_frequency, vi.fperiod, vi.alpha, 1.0, tr[1][:, 0], tr[0], tr[2])

    th = [threading.Thread(target=work) for _ in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    total = nsamp * n_utts_per_thread * cores
    return {
        "value": total / dt, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"{n_utts_per_thread * cores} utterances of {nsamp} samples (same synthetic "
                  f"utterance as the GPU batch), one per thread x {n_utts_per_thread}, "
                  f"{dt:.2f} s wall; C restatement of jbonsai's CPU path (oracle/), gcc -O2",
    }


def pcm_slab_tensor(batch):
    """The batch's PCM slab (jb_batch_device_pcm) as a zero-copy torch tensor on its device (f64,
    or i16 for a pcm_i16 batch), for an RCCL gather by the caller.  The tensor aliases
    library-owned memory: valid until the batch is closed, contents valid after sync().  Lives
    here, not in the package: the product does not import torch."""
    import torch

    p, n = batch.device_pcm()
    i16 = bool(batch.flags & 64)  # JB_BATCH_PCM_I16

    class _Slab:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i2" if i16 else "<f8",
                                    "data": (int(p), False), "version": 2}

    dev = batch.device if batch.device >= 0 else torch.cuda.current_device()
    return torch.as_tensor(_Slab(), device=torch.device("cuda", dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=0, help="frames per utterance (0 = 25,546 = ~128 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cu-split", type=int, default=-1,
                    help="CUs per XCD (of 32) given to parameter generation when two batches are in "
                         "flight; the vocoder gets the rest (jb_batch_opts.mlpg_cus_per_xcd); "
                         "-1 = default (0 with --pipeline 1)")
    ap.add_argument("--distinct", type=int, default=1,
                    help="number of DISTINCT synthetic utterances tiled over the batch (default 1 = BASELINE "
                         "config 2's copies of one utterance); >1 shows the cost of real hand-off failures")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE config 3's per-GPU share instead of config 2: --batch distinct utterances "
                         "of seed-fixed lengths U[400, 25546] frames (synth.mixed_lengths); not the headline line")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed steps, gather every rank's PCM slab to rank 0 with RCCL "
                         "(torch.distributed gather over xGMI) and report its time as gather_ms; "
                         "never part of `value` (SURVEY 8e)")
    ap.add_argument("--beta", type=float, default=0.0,
                    help="post-filter coefficient (off-config: BASELINE's metric is quoted at beta = 0)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="batches in flight per GPU (2: one batch's parameter generation overlaps the "
                         "other's vocoder on separate HIP streams; 1: strictly one step at a time)")
    args = ap.parse_args()

    import torch  # first: so that this process uses ONE HIP runtime (same SONAME as ours)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # JB_BENCH_REHEARSE=1: rehearsal of the multi-rank path on a ONE-GPU box -- every rank uses
    # device 0 and the collectives run over gloo on host tensors (RCCL refuses two ranks on one
    # device).  Numbers from such a run mean nothing; it only exercises the code path.
    rehearse = os.environ.get("JB_BENCH_REHEARSE", "0") != "0"
    if rehearse:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)

    import jbonsai_amd as J
    from jbonsai_amd import synth

    eng = J.Engine.load([VOICE])
    tab = synth.VoiceTables(eng)
    vi = eng.voice_info()
    vi.beta = args.beta
    frames = args.frames or synth.T_128S
    # every utterance of the batch is the same sequence (BASELINE config 2: "256 copies"),
    # uploaded once and aliased; outputs / workspace / filter state are per utterance
    utt = synth.synth_utterance(tab, frames, 0)
    nd = max(1, min(args.distinct, args.batch))
    utts = [utt] + [synth.synth_utterance(tab, frames, 1000 + i) for i in range(1, nd)]
    depth = max(1, args.pipeline)
    cu_split = args.cu_split if args.cu_split >= 0 else (DEFAULT_CU_SPLIT if depth > 1 else 0)
    if args.mixed:
        lens = synth.mixed_lengths(args.batch, seed=3 + rank)
        batch_utts = [synth.synth_utterance(tab, T, 2000 + i) for i, T in enumerate(lens)]
    else:
        batch_utts = [utts[i % nd] for i in range(args.batch)]
    batches = [J.Batch(vi, batch_utts, device=local_rank, mlpg_cus_per_xcd=cu_split) for _ in range(depth)]
    batch = batches[0]
    samples_per_step = batch.total_samples

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 0)):
        for b_ in batches:
            b_.run()
        for b_ in batches:
            b_.sync()
    barrier()
    t0 = time.perf_counter()
    voc_ms = []
    inflight = [False] * depth
    for k in range(args.steps):
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
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    gather_ms = None
    if args.gather and dist is not None and not rehearse:
        # optional sink of north_star: PCM of all ranks on GPU 0.  The slab is library-owned device
        # memory viewed zero-copy; rank 0 needs world x 12.6 GB of HBM for config 2.
        slab = pcm_slab_tensor(batch)
        dest = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
        barrier()
        tg = time.perf_counter()
        dist.gather(slab, dest, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        del dest

    if rank == 0:
        total = samples_per_step * world * args.steps
        value = total / dt
        voc_avg_ms = sum(voc_ms) / len(voc_ms)
        achieved = B_ALG * samples_per_step / (voc_avg_ms * 1e-3) / 1e9
        traffic = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                tj = json.loads(tf.read_text())
                if tj.get("batch") == args.batch and tj.get("frames") == frames:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        info = batch.info()
        out = {
            "metric": "48 kHz PCM samples/sec (whole node), batched utterances",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
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
                "samples_per_step_per_gpu": samples_per_step, "parallelism": f"utterance-sharded x{world}",
                "batches_in_flight": depth, "mlpg_cus_per_xcd": cu_split, "distinct_utterances": nd,
                "beta": args.beta,
                "chunks_settled_at_checkpoint_last_step": batch.redo_stats()[0],
                "vocoder_chunk_frames": info["chunk_frames"], "vocoder_warmup_frames": info["warmup_frames"],
                "vocoder_work_items": info["n_items"], "chunks_redone_last_step": info["n_redo"],
            },
            "realtime_factor": value / vi.sampling_frequency,
            **({"gather_ms": gather_ms} if gather_ms is not None else {}),
            "roofline": {
                "bound": "hbm", "kernel": "k_vocoder_lt" if info["chunk_frames"] and info["n_items"] >= 16384 else "k_vocoder", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel_ms": voc_avg_ms, "alg_bytes_per_sample": B_ALG,
                "note": "recursive IIR: FP64 VALU-issue bound, not HBM bound (DESIGN.md section 4); see valu_f64",
                # the bound that actually binds: useful f64 flops of the path (1.39 kflop per
                # output sample) over the same kernel time, against the FP64 vector peak of the guide
                "valu_f64": {"achieved_tflops": FLOP_PER_SAMPLE * samples_per_step / (voc_avg_ms * 1e-3) / 1e12,
                             "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                             "frac": FLOP_PER_SAMPLE * samples_per_step / (voc_avg_ms * 1e-3) / 1e12
                                     / FP64_VALU_PEAK_TFLOPS},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(utt, vi)
        print(json.dumps(out), flush=True)
    for b_ in batches:
        b_.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
