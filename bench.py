#!/usr/bin/env python3
"""bench.py -- headline benchmark: end-to-end frame-pairs/s (V2ce3d UNet + LDATI) at 346x260.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload e2e|pano|ldati_stress|ldati_sparse|voxelize]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    (`python bench.py --gpus N` without torchrun starts the N ranks itself, as child processes)

A "step" is one pass of the hot path over one batch resident in HBM: `--batch` (default 4, the
BASELINE.json configs[1] "batch=4 sliding frame-pairs" case) 16-pair sequences of preprocessed
346x260 frames -> V2ce3d -> LDATI (count, emit with the per-frame offset fused, packed 13-byte
records) on the device.  The loop is software-pipelined exactly like the product driver
(pipeline.run_clip): LDATI's emit phase of step k-1 is enqueued behind the model of step k, so the
host's read of the segment table never drains the GPU -- and (round 6) LDATI has its own high-priority
stream, so its workgroups run beside the next step's convolutions, on the CUs a persistent conv launch
has no tile left for (V2CE_LDATI_STREAM=0: on the main stream); all K steps complete inside the timed region.
With N > 1 every rank processes its own contiguous block of sequences (weak scaling, no data-path
collective) and the packed events are gathered to rank 0 over RCCL inside the timed region.
`--workload pano` is BASELINE config 4 (1384x260, batch 8): one 346-wide tile per GPU of a 4-rank
group, all-to-all re-shard from W-tiles to frame-pairs, full-width LDATI, gather.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from v2ce_toolbox_amd import synth                                   # noqa: E402
from v2ce_toolbox_amd.LDATI import ldati_begin, ldati_device         # noqa: E402
from v2ce_toolbox_amd.v2ce_3d import V2ce3d                          # noqa: E402
from v2ce_toolbox_amd import dist as vdist                           # noqa: E402
from v2ce_toolbox_amd import glue                                    # noqa: E402

H, W, SEQ = 260, 346, 16
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_F16_MATRIX_TFLOPS = 16 * PEAK_F32_MATRIX_TFLOPS   # same table: f32 MFMA = 1/16 of the BF16/F16 rate (~2.5 PF dense)
# split-half kernels execute 3 fp16 MFMAs per algorithmic product (hi*hi + hi*lo + lo*hi)
PEAK_SPLIT_TFLOPS = PEAK_F16_MATRIX_TFLOPS / 3
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
FLOP_PER_PAIR = 135.58e9
DTYPE_NOTE = {"f32": "f32", "f16x2": "f32-as-f16x2 (operands split hi/lo in fp16, 3 MFMAs per product, f32 accumulate)"}


def make_inputs(batch, first_seq, device):
    """[batch,16,2,H,W] f32 preprocessed frame pairs (v2ce.py:45-64) of synthetic drifting frames."""
    xs = []
    for s in range(batch):
        fr = synth.synthetic_frames(SEQ + 1, H, W, seed=1000 + first_seq + s)
        xs.append(glue.image_pre_processing(fr))
    return torch.from_numpy(np.stack(xs)).to(device)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline():
    """The oracle (CPU restatement) on a bounded sample of the same workloads, as BASELINE.md 3 plans:
    config C1 (one 17-frame 346x260 sequence = 16 frame-pairs: torch-CPU stage 1 on the host cores +
    scalar C LDATI) and a slice of one C5 stress chunk through the C LDATI oracle."""
    from oracle import ldati as O
    from oracle import unet as U
    cores = os.cpu_count() or 1
    threads = min(64, cores)                     # more threads than this slow the oneDNN convs down
    torch.set_num_threads(threads)
    sd = synth.make_state_dict(0)
    fr = synth.synthetic_frames(SEQ + 1, H, W, seed=1000)
    x = torch.from_numpy(glue.image_pre_processing(fr)[None])
    t0 = time.perf_counter()
    vox = U.forward(sd, x).contiguous().numpy().reshape(SEQ, 2, 10, H, W)
    t1 = time.perf_counter()
    seg, ts, _, _, _ = O.emit_soa(vox, fps=30, seed=1)
    t2 = time.perf_counter()
    c5_pairs = 4
    vs = synth.synthetic_voxels(c5_pairs, H, W, seed=7, regime="stress")
    t3 = time.perf_counter()
    seg5, _, _, _, _ = O.emit_soa(vs, fps=30, seed=1)
    t4 = time.perf_counter()
    return {"value": SEQ / (t2 - t0), "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "cpu": f"{cpu_model()} ({cores} logical cores on the box)",
            "sample": f"C1: one 346x260 sequence (16 frame-pairs): oracle stage 1 on torch-CPU {t1 - t0:.1f} s "
                      f"({threads} threads) + scalar C LDATI {t2 - t1:.1f} s (1 thread), {int(seg.sum())} events",
            "c5_ldati": {"value": int(seg5.sum()) / (t4 - t3) / 1e6, "unit": "Mevents/s", "cores": 1,
                         "sample": f"{c5_pairs} of the 24 frame-pairs of the C5 stress chunk (6*U[0,1) voxels), "
                                   f"{int(seg5.sum())} events, scalar C oracle {t4 - t3:.1f} s"}}


def bench_voxelize(args, device, rank, world):
    """events of 24 stress frame-pairs (one LDATI chunk) -> one [20,260,346] volume per
    frame-pair.  A step = the 24 voxelise calls; HBM-bound: 13 B read per event + the zeroed volume."""
    from v2ce_toolbox_amd.voxelize import gen_discretized_event_volume
    pairs = 24
    vox = torch.from_numpy(synth.synthetic_voxels(pairs, H, W, seed=7 + rank, regime="stress")).to(device)
    ev = ldati_device(vox, fps=30, seed=0x5EED)
    ends = np.cumsum(ev.frame_counts)
    frames = [(ev.ts[e - c:e].contiguous(), ev.x[e - c:e].contiguous(), ev.y[e - c:e].contiguous(),
               ev.p[e - c:e].contiguous()) for e, c in zip(ends, ev.frame_counts)]
    n_events = int(ends[-1])

    def step():
        for f in frames:
            gen_discretized_event_volume(f, (20, H, W))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_s = e0.elapsed_time(e1) * 1e-3
    nbytes = 13.0 * n_events + pairs * 20 * H * W * 4.0          # events read once + volumes written once
    achieved = nbytes * args.steps / gpu_s / 1e9
    line = {"metric": "frame-pairs/sec voxelised (events -> [20,260,346] volume)", "value": pairs * args.steps / dt,
            "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "voxeliser only: LDATI events of 24 stress frame-pairs (6*U[0,1) voxels), one call per frame-pair",
                       "events_per_step": n_events},
            "mevents_per_s": n_events * args.steps / dt / 1e6,
            "roofline": {"bound": "hbm", "kernel": "v2ce_voxelize_events (time_range + voxelize kernels, memset)",
                         "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "traffic": None, "bytes_per_step": nbytes}}
    if not args.no_cpu_baseline:
        from oracle.voxelize import gen_discretized_event_volume as oracle_vox
        rec = ev.to_recarrays()[0]
        t0 = time.perf_counter()
        oracle_vox(rec, (20, H, W))
        t1 = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": 1.0 / t1, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                "sample": f"1 frame-pair, {len(rec)} events, numpy restatement"}
    if rank == 0:
        print(json.dumps(line))


def self_launch(n_gpus: int) -> int:
    """Run this script under torch.distributed.run with `n_gpus` ranks in a child process."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                 # (the ranks are started as CHILD processes, never exec'd over this one)
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes on this driver)
    return subprocess.call(cmd, env=env)


def fresh_model(precision, device):
    # guard='deferred': like the product driver (glue.run_guarded around pipeline.run_clip) the bench checks the
    # range guard once after the timed steps (`range_guard` in the JSON line) instead of once per call
    m = V2ce3d(precision=precision, guard="deferred")
    m.load_state_dict(synth.make_state_dict(0))
    return m.eval().to(device)


def host_to_host(args, device, steps, comm=None, world=1):
    """SURVEY 8d's end-to-end INCLUDING the host hand-over, through the product driver
    (pipeline.run_clip): u8 frames in host memory -> packed events in host memory, for a clip of
    steps x batch sequences per GPU.  H2D / compute / D2H overlap on three streams.  With more than one rank the
    clip is world x as long and every reference batch (world x batch sequences) is shared out over the ranks, exactly
    like the CLI under torchrun; the records reach host memory the way `V2CE_GATHER` says (default 'device': RCCL gather
    to rank 0's HBM + rank 0's PCIe link into its pinned sink; 'host': every rank over its own link into one shared segment)."""
    from v2ce_toolbox_amd import pipeline
    n_seq = steps * args.batch * world
    base = synth.synthetic_frames(SEQ + 1, H, W, seed=1000)
    frames = np.concatenate([base[:SEQ]] * n_seq + [base[SEQ:SEQ + 1]])          # n_seq*16 + 1 frames
    model = fresh_model(args.precision, device)
    kw = dict(infer_type="center", batch_size=args.batch * world, fps=30, seed=0x5EED, device=str(device), reuse_output=True)
    if comm is not None:
        kw["comm"] = comm
    # warm-up clip of the same length: device / pinned allocators, and the page-locked output buffer,
    # which a process keeps between clips (reuse_output; locking 1.9 GB costs ~140 ms once)
    pipeline.run_clip(frames, model, **kw)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    trace = {} if os.environ.get("V2CE_TRACE") else None
    t0 = time.perf_counter()
    ev = pipeline.run_clip(frames, model, trace=trace, **kw)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
        if ev is None:                                                          # only rank 0 holds the clip's events
            return None
    if trace is not None:
        print("host_to_host trace (s):", {k: (round(v, 4) if not isinstance(v, list) else v) for k, v in trace.items()},
              "total", round(dt, 4), file=sys.stderr)
    pairs = n_seq * SEQ
    # the box's own device -> pinned-host copy rate for one step's records: on boxes where it is below
    # d2h_bytes_per_step / GPU step time, the copy-out stream (not the GPU) paces the pipeline
    nbytes = int(len(ev) * 13 / steps / world)                                  # one rank's records of one step
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    best = float("inf")
    for _ in range(4):
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - c0)
    return {"value": pairs / dt, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "n_gpus": world,
            "gather": vdist.default_gather_mode(world) if world > 1 else None,
            "mevents_per_s": len(ev) / dt / 1e6, "h2d_bytes_per_step": int(args.batch * (SEQ + 1) * H * W),
            "d2h_bytes_per_step": nbytes, "d2h_copy_gb_per_s": nbytes / best / 1e9, "d2h_copy_ms_per_step": 1e3 * best,
            "what": "u8 frames in host memory -> event_stream array in pinned host memory through pipeline.run_clip "
                    "(copy-in / compute / copy-out streams); second clip of the process: the page-locked output "
                    "buffer is reused (the first clip additionally pays ~70 us per MB to lock it)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # (0.3 s of GPU time; five steps sat 1-2 % below the steady state)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="16-pair sequences per step per GPU (default 4; pano 8)")
    ap.add_argument("--workload", default="e2e", choices=["e2e", "pano", "ldati_stress", "ldati_sparse", "voxelize"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f16x2", choices=["f32", "f16x2"],
                    help="stage-1 3x3x3 conv arithmetic: f16x2 = f32 operands split into two fp16 halves, 3 fp16 "
                         "MFMAs per product, f32 accumulation (error vs f64 equals the exact path's: "
                         "profiles/r01_d_precision_report.json); f32 = exact f32 MFMA")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the extra exact-f32 measurement")
    ap.add_argument("--no-host-to-host", action="store_true", help="skip the host-to-host (PCIe-inclusive) measurement")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 8 if args.workload == "pano" else 4

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (one per
        # GPU, RCCL rendezvous on 127.0.0.1) before this process has made any GPU call, and exit
        # with their code.  Rank 0 of the children prints the JSON line.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    # V2CE_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, per-step EventGather, barriers, reductions)
    # with a world of one -- how that path is exercised on a 1-GPU box (under torchrun --nproc-per-node 1)
    dist_on = world > 1 or os.environ.get("V2CE_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.workload == "voxelize":
        return bench_voxelize(args, device, rank, world)

    b, fps = args.batch, 30
    model, x, vox_fixed = None, None, None
    pano_tiles = 4
    tile_parallel = args.workload == "pano" and world % pano_tiles == 0
    if args.workload == "e2e":
        pairs_per_rank = b * SEQ
        first_pair = rank * pairs_per_rank                       # contiguous block of sequences per rank
        model = fresh_model(args.precision, device)
        x = make_inputs(b, rank * b, device)
    elif args.workload == "pano":
        # 1384x260 = four 346-wide tiles (v2ce.py:103-126 makes one model call per tile).  A group of
        # four ranks shares a batch (one tile each); other world sizes give every rank its own batch
        # and run the four tiles one after the other like the reference.
        grp_index, tile_index = divmod(rank, pano_tiles) if tile_parallel else (rank, None)
        grp = vdist.subgroup(pano_tiles, rank, world) if tile_parallel else None      # RCCL group of the four tile ranks
        model = fresh_model(args.precision, device)
        xs = [make_inputs(b, (grp_index * pano_tiles + t) * b, device)
              for t in ([tile_index] if tile_parallel else range(pano_tiles))]
        pairs_per_rank = b * SEQ // pano_tiles if tile_parallel else b * SEQ      # full-width pairs LDATI sees
        first_pair = grp_index * b * SEQ
    else:
        regime = "stress" if args.workload == "ldati_stress" else "sparse"
        pairs_per_rank = 24                                   # reference stage-2 chunk (v2ce.py:302)
        first_pair = rank * pairs_per_rank
        vox_fixed = torch.from_numpy(synth.synthetic_voxels(pairs_per_rank, H, W, seed=7 + rank, regime=regime)).to(device)
    # per-frame offsets int(i*1/fps*1e6) (v2ce.py:365) of every global frame-pair index, uploaded once
    offsets = torch.tensor([glue.frame_offset_us(i, fps) for i in range((world + 1) * max(b, 8) * SEQ)],
                           dtype=torch.int64).to(device)
    ldati_prof, conv_prof = [], []
    n_events = [0]
    # rank 0 receives every rank's records of a step over RCCL on a communication stream (dist.StreamedGather, the
    # product driver's: pipeline.run_clip); the byte counts are read one step later, so no rank waits on its compute stream
    comm = vdist.default_comm(force=dist_on)
    gather_mode = vdist.default_gather_mode(world)        # the product's default: 'device' (V2CE_GATHER=host opts in)
    d2h = {"bytes": 0, "s": 0.0}

    def new_exchange():
        """The exchange pipeline.run_clip builds for a clip: 'host' = every rank downloads its own records into its slice of
        one shared host segment (only byte counts cross RCCL); 'device' = RCCL gather to rank 0's HBM, rank 0 downloads
        everything into its pinned sink.  The bench's segment is a 4 GiB ring (a long run must not fill /dev/shm)."""
        if not dist_on:
            return None, None
        from v2ce_toolbox_amd import pipeline
        if gather_mode == "host":
            ring = 4 << 30
            if not host_segments:
                # one segment per process, made by the warm-up run and reused by the timed one: every rank maps and page-locks the
                # ring once (~70 us per MB: 0.3 s for 4 GiB -- a per-clip cost in the product) and then writes it by DMA
                use_reg = os.environ.get("V2CE_HOST_SEGMENT_MB") != "0"
                path = None
                if rank == 0:
                    path = pipeline._shared_segment_path()
                    with open(path, "wb") as f:
                        if use_reg:
                            f.truncate(ring)
                path = comm.broadcast_object(path, src=0)
                host_segments.append((path, vdist.RegisteredSegment(path, ring) if use_reg else None))
                if rank == 0:
                    stale_segments.append(path)
            path, seg = host_segments[0]
            ex = vdist.HostDirectGather(comm, device, path, 0, False, segment=seg)
            ex.ring_bytes = ring
            return ex, path
        sink = pipeline.EventSink(device, 1, reuse=True) if rank == 0 else None

        def on_pieces(pieces, stream):
            for p in pieces:
                sink.push(p, 0, src_stream=stream)
            sink.used = 0                                       # (the bench keeps one step's worth: the sink is a staging area here)
            return sink.last_done
        return comm.streamed_gather(on_pieces if rank == 0 else None, dst=0), sink
    gather, gather_aux = None, None
    stale_segments, host_segments = [], []

    def front(profile):
        """Stage 1 + LDATI count of one step; returns the pending LDATI call."""
        fp = first_pair
        if args.workload == "e2e":
            model.profile = [] if profile else None
            vox = model(x).view(pairs_per_rank, 2, 10, H, W)
            conv_prof.extend(model.profile or [])
        elif args.workload == "pano":
            model.profile = [] if profile else None
            if tile_parallel:
                part = model(xs[0]).view(b * SEQ, 2, 10, H, W)
                vox, lo = vdist.tiles_to_pairs(part, [W] * pano_tiles, tile_index, grp)
                fp = first_pair + lo
            else:
                vox = torch.cat([model(xt).view(b * SEQ, 2, 10, H, W) for xt in xs], dim=-1)
            conv_prof.extend(model.profile or [])
        else:
            vox = vox_fixed
        add = offsets[fp:fp + vox.shape[0]] if model is not None else None
        if side_stream is not None:                             # LDATI beside the next step's convs, as pipeline.run_clip runs it
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(ready)
                h = ldati_begin(vox, fps=fps, seed=0x5EED, frame_base=fp, frame_ts_add=add, profile=ldati_prof if profile else None)
            vox.record_stream(side_stream)
            return h
        return ldati_begin(vox, fps=fps, seed=0x5EED, frame_base=fp, frame_ts_add=add,
                           profile=ldati_prof if profile else None)

    # stage 2 on its own stream (pipeline.run_clip; V2CE_LDATI_STREAM=0: behind the model on the main stream)
    side_stream = torch.cuda.Stream(device=device, priority=int(os.environ.get("V2CE_LDATI_STREAM_PRIORITY", "-1"))) if os.environ.get("V2CE_LDATI_STREAM", "1") != "0" else None

    def back(pending):
        if side_stream is not None:
            with torch.cuda.stream(side_stream):
                ev = pending.finish()
                packed = ev.packed()
            n_events[0] += ev.num_events
            if gather is not None:                              # the exchange reads the records on the main stream
                emitted = torch.cuda.Event()
                emitted.record(side_stream)
                torch.cuda.current_stream(device).wait_event(emitted)
                packed.record_stream(torch.cuda.current_stream(device))
                gather.submit(packed)
            return packed, ev
        ev = pending.finish()
        packed = ev.packed()
        n_events[0] += ev.num_events
        if gather is not None:
            gather.submit(packed)
        return packed, ev

    status = [None]

    def run_steps(k, profile):
        pending = None

        def done(p):
            _, ev = back(p)
            if ev._status is not None:                          # LDATI's device status word, folded on the stream
                with torch.cuda.stream(side_stream) if side_stream is not None else contextlib.nullcontext():
                    status[0] = ev._status.clone() if status[0] is None else torch.maximum(status[0], ev._status)
        nonlocal gather, gather_aux
        t_x = time.perf_counter()
        gather, gather_aux = new_exchange()
        t_new = time.perf_counter() - t_x
        for _ in range(k):
            nxt = front(profile)
            if pending is not None:
                done(pending)
            pending = nxt
        if pending is not None:
            done(pending)
        t_loop = time.perf_counter() - t_x
        if gather is not None:
            gather.drain()                                      # the last step's exchange
            if gather_mode == "host":
                gather.finalize()                               # every rank's records are in the shared segment (unlinked behind the
                                                                # timed region: the product hands the pages on as the array)
            elif rank == 0 and gather_aux.stream is not None:
                gather_aux.stream.synchronize()                 # rank 0's download of the last pieces
            if os.environ.get("V2CE_TRACE"):
                print(f"bench exchange: new {1e3 * t_new:.1f} ms, loop {1e3 * t_loop:.1f} ms, total {1e3 * (time.perf_counter() - t_x):.1f} ms", file=sys.stderr)

    run_steps(args.warmup, True)            # same code path as the timed steps (warms the HIP event pool too)
    # Per-launch HIP events around EVERY conv launch cost 0.13-0.24 ms of a 16 ms step (~90 event records): the warm-up steps carry
    # them all (the `kernels` table, the flop sums), the timed steps only those of the launch family with the most time -- the kernel
    # the `roofline` object is about, measured live over the timed region.  (--warmup 0: every launch is profiled in the timed steps.)
    warm_prof = []
    if model is not None and conv_prof and args.warmup >= 1 and not os.environ.get("V2CE_BENCH_PROFILE_ALL"):
        torch.cuda.synchronize()
        n_fw = len(model.profile or [])                           # launches of one forward call (the last one's list)
        calls = len(conv_prof) // n_fw if n_fw else 0
        if n_fw and calls * n_fw == len(conv_prof):
            keep = conv_prof[n_fw:] if calls > 1 else conv_prof  # (the first call of a run pays the one-off set-up: left out when there are others)
            warm_prof = [(name, flops, e0.elapsed_time(e1) * 1e-3, executed) for name, flops, e0, e1, executed in keep]
            tot = {}
            for name, _, t, _ in warm_prof:
                tot[name] = tot.get(name, 0.0) + t
            dom = max(tot, key=tot.get)
            order = [p[0] for p in conv_prof[-n_fw:]]
            model.profile_filter = {i for i, nme in enumerate(order) if nme == dom}
    ldati_prof.clear()
    conv_prof.clear()
    n_events[0] = 0
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # First-hardware-curve readiness: from eight ranks on the same steps are timed once more with the OTHER exchange,
    # so that the first run on an 8-GPU node shows north_star's RCCL gather ('device', the default and `value`) and the per-rank
    # PCIe download ('host') side by side.  V2CE_BENCH_BOTH_GATHERS=1 forces it at any world > 1, =0 switches it off.
    main_gather, main_mode, other_gather, main_events = gather, gather_mode, None, n_events[0]
    both = os.environ.get("V2CE_BENCH_BOTH_GATHERS")
    if dist_on and args.workload == "e2e" and both != "0" and (world >= 8 or both == "1"):
        gather_mode = "host" if main_mode == "device" else "device"
        try:
            model.profile_filter = set()                      # no per-launch events in this run
            run_steps(1, False)
            torch.cuda.synchronize()
            torch.distributed.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.steps, False)
            torch.cuda.synchronize()
            torch.distributed.barrier()
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            other_gather = {"mode": gather_mode, "ms_per_step": 1e3 * float(tt[0]) / args.steps,
                            "value": world * pairs_per_rank * args.steps / float(tt[0]), "unit": "frame-pairs/s",
                            "gathered_bytes_per_step": gather.bytes_last}
            if gather_mode == "host":
                other_gather["registered_segment"] = bool(host_segments and host_segments[0][1] is not None)
        except Exception as e:                                # noqa: BLE001 -- the main line must still be printed
            other_gather = {"mode": gather_mode, "error": f"{type(e).__name__}: {e}"}
        gather, gather_mode, n_events[0] = main_gather, main_mode, main_events
    for _, rseg in host_segments:
        if rseg is not None:
            rseg.close()
    for seg in stale_segments:
        if os.path.exists(seg):
            os.unlink(seg)
    if status[0] is not None and int(status[0].item()) != 0:
        sys.exit("bench.py: LDATI reported a segment it could not order (device status word)")
    events = float(n_events[0])
    if dist_on:
        t = torch.tensor([dt, events], dtype=torch.float64, device=device)
        tmax = t.clone()
        torch.distributed.all_reduce(tmax[:1], op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(t[1:], op=torch.distributed.ReduceOp.SUM)
        dt, events = float(tmax[0]), float(t[1])
    total_pairs = world * pairs_per_rank * args.steps            # pano: full-width (1384-column) frame-pairs

    # ---- per-kernel HIP-event timings: the dominant family's from the timed region, the others' from the warm-up steps
    per, timed_names = {}, set()
    for name, flops, e0, e1, executed in conv_prof:
        d = per.setdefault(name, [0.0, 0.0, 0, 0.0])
        d[0] += e0.elapsed_time(e1) * 1e-3
        d[1] += flops                   # algorithmic: the reference convolution's multiply-adds
        d[2] += 1
        d[3] += executed                # what the launch multiplies (less for the phase-folded decoder kernels)
        timed_names.add(name)
    from_warm = set()
    if warm_prof and getattr(model, "profile_filter", None):
        n_dom = len(model.profile_filter)
        timed_calls = sum(v[2] for v in per.values()) / n_dom             # forward calls the timed events cover
        warm_calls = sum(1 for p in warm_prof if p[0] in timed_names) / n_dom
        k = timed_calls / warm_calls if warm_calls else 1.0               # warm-up totals scaled to the timed region's number of calls:
        for name, flops, t, executed in warm_prof:                       # sums over the table are sums over the timed steps
            if name in timed_names:
                continue
            d = per.setdefault(name, [0.0, 0.0, 0.0, 0.0])
            d[0] += t * k
            d[1] += flops * k
            d[2] += k
            d[3] += executed * k
            from_warm.add(name)
    kernels = {k: {"launches": int(round(v[2])), "avg_ms": 1e3 * v[0] / v[2], "tflops": v[1] / v[0] / 1e12,
                   **({"executed_tflops": v[3] / v[0] / 1e12} if v[3] != v[1] else {}),
                   **({"events": "warm-up steps"} if k in from_warm else {})}
               for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])}
    # LDATI = count kernels + emit kernels (the host read of the segment table between them is not GPU time)
    em = [(e0.elapsed_time(e1) * 1e-3, nb) for tag, e0, e1, nb in ldati_prof if tag == "emit"]
    cnt_t = sum(e0.elapsed_time(e1) * 1e-3 for tag, e0, e1, nb in ldati_prof if tag == "count")
    em_t, em_bytes, em_n = sum(t for t, _ in em) + cnt_t, sum(nb for _, nb in em), len(em)
    ldati = None
    if em_n:
        ldati = {"kernel": "v2ce_ldati_count + v2ce_ldati_emit", "avg_ms": 1e3 * em_t / em_n,
                 "count_ms": 1e3 * cnt_t / em_n,
                 "achieved_GBps": em_bytes / em_t / 1e9, "frac_hbm_peak": em_bytes / em_t / 1e9 / PEAK_HBM_GBS,
                 "algorithmic_bytes_per_launch": em_bytes / em_n}
        if model is not None and side_stream is not None:
            ldati["note"] = ("elapsed on LDATI's own stream BESIDE the next step's convolutions (pipeline.run_clip): its workgroups run where "
                             "a persistent conv launch has no tile left for a CU, so this is a stretched duration, not the kernels' cost -- "
                             "alone on the chip the same call takes 0.40 ms (V2CE_LDATI_STREAM=0; the ldati_* workloads)")
    if model is not None:
        name = max(per, key=lambda k: per[k][0])
        v = per[name]
        split = "f16x2" in name or "up_kernel" in name or "wt_kernel" in name
        peak = PEAK_SPLIT_TFLOPS if split else PEAK_F32_MATRIX_TFLOPS
        ws3 = [q[3] / q[0] / 1e12 / PEAK_SPLIT_TFLOPS for k, q in per.items() if "ws_kernel<3," in k or "up_kernel<" in k or "wt_kernel<" in k]
        roofline = {"bound": "mfma", "kernel": name, "achieved": v[1] / v[0] / 1e12,
                    "peak": peak, "unit": "TFLOP/s",
                    "frac": v[1] / v[0] / 1e12 / peak, "traffic": None,
                    "avg_launch_ms": 1e3 * v[0] / v[2], "flop_per_launch": v[1] / v[2],
                    "executed_flop_per_launch": v[3] / v[2], "executed_frac": v[3] / v[0] / 1e12 / peak,
                    "peak_note": ("algorithmic (f32-equivalent) FLOP; the kernel executes 3 fp16 MFMAs per product, so "
                                  f"peak = dense fp16 MFMA peak {PEAK_F16_MATRIX_TFLOPS:.0f} / 3") if split else
                                 "dense f32 MFMA peak",
                    "all_conv_tflops": sum(q[1] for q in per.values()) / sum(q[0] for q in per.values()) / 1e12,
                    "min_3x3x3_split_variant_frac": min(ws3) if ws3 else None,
                    "frac_note": "achieved / frac count the ALGORITHMIC flop of the reference's convolution (SURVEY 8d: 135.58 GFLOP per "
                                 "frame-pair); executed_* count what the launch really multiplies (the decoder conv1 kernels fold the "
                                 "2x-upsampled channels' 27 taps into 12: DESIGN 4.1e; the Winograd-T kernels multiply 36 instead of 54 "
                                 "times per pair of time steps: DESIGN 4.1g); min_3x3x3_split_variant_frac is on executed flop"}
    else:
        roofline = {"bound": "hbm", "kernel": "v2ce_ldati_count + v2ce_ldati_emit (count_tiles, tile_scan, tile_pass, bucket_scan, bucket_sort)",
                    "achieved": ldati["achieved_GBps"],
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ldati["frac_hbm_peak"], "traffic": None,
                    "avg_launch_ms": ldati["avg_ms"], "bytes_per_launch": ldati["algorithmic_bytes_per_launch"]}

    # HBM-side traffic of the same kernel(s) from the latest committed PMC summary of this workload
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_summary.py): bytes per
    # launch, or null when no summary for this workload is committed
    from v2ce_toolbox_amd import hip as _hip
    my_hash = _hip.source_hash()

    def fresh(path):
        """A committed counter summary counts only when it was collected from THIS tree's kernel sources."""
        data = json.load(open(path))
        prov = data.pop("_provenance", None) or {}
        if prov.get("source_hash") != my_hash:
            raise LookupError(f"{os.path.basename(path)} was collected from other kernel sources ({prov.get('source_hash')} != {my_hash})")
        return data, prov
    def counter_key(table):
        """The counter tables carry rocprofv3's demangled names (a bool template argument prints as true / false)."""
        k = roofline["kernel"].replace(" ", "")
        if k in table:
            return k
        for a, b in ((",1>", ",true>"), (",0>", ",false>")):
            if k.endswith(a) and k[:-len(a)] + b in table:
                return k[:-len(a)] + b
        return k

    stale = []
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_pmc_traffic.json")))
        if not files:
            raise LookupError(f"no counter summary committed for this workload (profiles/r*_{args.workload}_pmc_traffic.json)")
        pmc, prov = fresh(files[-1])
        if model is not None:
            roofline["traffic"] = pmc[counter_key(pmc)]["traffic_bytes"]
        else:   # all ldati_* kernels of one steady-state call: one-time device checks and the kernels of the first call only
            # (a fused pass whose expectation missed, its two-pass repeat: at most half the launches of the others) left out
            ld = {k: v for k, v in pmc.items() if k.startswith("ldati_") and not any(x in k for x in ("check", "probe", "slope_tab", "commit"))}
            nmax = max(v["launches"] for v in ld.values())
            roofline["traffic"] = sum(v["traffic_bytes"] for v in ld.values() if 2 * v["launches"] > nmax)
        roofline["traffic_source"] = os.path.basename(files[-1]) + f" (PMC FETCH_SIZE x2 + WRITE_SIZE, avg per launch; {prov['lib_version']}, sources {prov['source_hash']})"
    except LookupError as e:
        stale.append(str(e))
    except Exception:
        pass
    # counter-derived shares from the latest committed SQ summaries (tools/profile_counters.sh -> tools/counters_summary.py):
    # the dominant conv kernel's MFMA-busy share of the SIMD cycles; LDATI's VALU-issue roofline fraction
    try:
        import glob
        if model is not None:
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_e2e_mfma_busy.json")))
            if not files:
                raise LookupError("no counter summary committed for this workload (profiles/r*_e2e_mfma_busy.json)")
            data, prov = fresh(files[-1])
            mb = data[counter_key(data)]
            roofline["mfma_busy"] = mb["mfma_busy"]
            roofline["mfma_busy_source"] = os.path.basename(files[-1]) + f" (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8), avg per launch; {prov['lib_version']}, sources {prov['source_hash']})"
        else:
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_sq_counters.json")))
            if not files:
                raise LookupError(f"no counter summary committed for this workload (profiles/r*_{args.workload}_sq_counters.json)")
            sq, prov = fresh(files[-1])
            roofline["valu_frac"] = sq["valu_frac"]
            roofline["valu_lane_slots_per_event"] = sq["valu_lane_slots_per_event"]
            roofline["valu_source"] = os.path.basename(files[-1]) + f" (SQ_INSTS_VALU x 64 / (1024 SIMDs x 16 lanes/clk x GRBM_GUI_ACTIVE / 8); {prov['lib_version']}, sources {prov['source_hash']})"
    except LookupError as e:
        stale.append(str(e))
    except Exception:
        pass
    if stale:
        roofline["counter_fields_dropped"] = stale      # (no counter-derived number of another build rides on this line)

    h2h_multi = None
    if world > 1 and args.workload == "e2e" and not args.no_host_to_host:
        h2h_multi = host_to_host(args, device, 4, comm=comm, world=world)        # 4 reference batches of world x batch sequences (N = 8: BASELINE config 3's clip)
    if rank == 0:
        workloads = {"e2e": f"346x260 center, batch={b} sequences x 16 frame-pairs per GPU, "
                            "V2ce3d (synthetic weights seed 0) + LDATI (Philox), inputs resident in HBM",
                     "pano": f"1384x260 pano (4 tiles of 346), batch={b} sequences x 16 frame-pairs per 4-GPU group: "
                             + ("one tile per GPU, all-to-all re-shard W-tiles -> frame-pairs, full-width LDATI"
                                if tile_parallel else "every rank runs the 4 tiles of its own batch serially (world not a multiple of 4)")
                             + "; inputs resident in HBM; value counts full-width frame-pairs",
                     "ldati_stress": "LDATI only, 24 frame-pairs of 6*U[0,1) voxels (C5 stress)",
                     "ldati_sparse": "LDATI only, 24 frame-pairs of relu(0.8*randn) voxels"}
        line = {
            "metric": "frame-pairs/sec end-to-end (UNet+LDATI), 346x260" if args.workload != "pano" else
                      "frame-pairs/sec end-to-end (UNet+LDATI), 1384x260 pano",
            "value": total_pairs / dt, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "value_is": "device-resident: inputs in HBM when the timed region starts, records left in HBM (the measurement contract of "
                        "this build: a PCIe-inclusive rate is never `value`); SURVEY 8d's own figure -- u8 frames in host memory -> "
                        "records in pinned host memory -- is `host_to_host.value` on the same line",
            "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "config": {"workload": workloads[args.workload], "frame_pairs_per_step_per_gpu": pairs_per_rank, "fps": fps,
                       "parallelism": (f"tile-per-GPU groups of 4 x {world // pano_tiles} over batches" if tile_parallel
                                       else f"dp{world} over sequences")},
            "mevents_per_s": events / dt / 1e6, "events_per_pair": events / total_pairs,
            "stage1_mfma_frac_e2e": ((FLOP_PER_PAIR * (pano_tiles if args.workload == "pano" else 1)) * total_pairs / dt / 1e12 / world /
                                     (PEAK_F32_MATRIX_TFLOPS if args.precision == "f32" else PEAK_SPLIT_TFLOPS))
            if model is not None else None,
            "roofline": roofline, "ldati": ldati, "kernels": kernels,
            "kernels_note": ("HIP events around every launch cost 0.13-0.24 ms of a step: the timed steps carry them on the launch family "
                             "with the most time only (the `roofline` kernel, measured live over the timed region); the other families' "
                             "rows come from the per-launch events of the warm-up steps, scaled to the timed region's number of calls")
            if any("events" in v for v in kernels.values()) else None,
        }
        if model is not None and per:
            n_fw = total_pairs / world        # frame-pairs this rank's profiled launches covered
            line["executed_flop_per_pair"] = sum(q[3] for q in per.values()) / n_fw
            line["algorithmic_flop_per_pair"] = sum(q[1] for q in per.values()) / n_fw
        if model is not None and args.precision == "f16x2":
            # the split-half range guard over the timed steps: worst per-launch bound vs its limit
            line["range_guard"] = {"worst_bound": model.range_guard_value(), "limit": model.RANGE_GUARD_LIMIT}
        if dist_on:
            line["gathered_bytes_per_step"] = gather.bytes_last
            line["rccl_world"] = world
            line["gather"] = {"mode": gather_mode,
                              "what": ("every rank downloads its own records over its own PCIe link into its slice of one shared host "
                                       "segment; RCCL carries one int64 byte count per rank and step (dist.HostDirectGather)") if gather_mode == "host"
                              else "RCCL gather of the padded record buffers to rank 0's HBM, rank 0 downloads them (dist.StreamedGather)",
                              "inside_timed_region": True}
            if gather_mode == "host":
                line["gather"]["registered_segment"] = bool(host_segments and host_segments[0][1] is not None)
                line["gather"]["dma_bytes_last_run"] = int(getattr(gather, "dma_bytes", 0))
            if other_gather is not None:
                line["gather"]["other_mode"] = other_gather       # the same steps timed with the other exchange (not `value`)
        if world == 1 and args.workload == "e2e":
            if not args.no_host_to_host:
                line["host_to_host"] = host_to_host(args, device, max(args.steps, 16))   # >= 16 batches (1025 frames): the drain of the last batch (0.5 ms LDATI + 152 MB D2H) is a fixed ~7 ms
            if args.precision != "f32" and not args.no_exact_f32:
                # the same step with exact f32 MFMA arithmetic in every conv, for reference (not `value`)
                model = None
                m32 = fresh_model("f32", device)

                def step32():
                    vox = m32(x).view(pairs_per_rank, 2, 10, H, W)
                    return ldati_device(vox, fps=fps, seed=0x5EED, frame_base=first_pair).packed()
                step32()
                torch.cuda.synchronize()
                t32 = time.perf_counter()
                for _ in range(3):
                    step32()
                torch.cuda.synchronize()
                t32 = (time.perf_counter() - t32) / 3
                line["exact_f32"] = {"value": pairs_per_rank / t32, "unit": "frame-pairs/s", "ms_per_step": 1e3 * t32, "steps": 3}
        if h2h_multi is not None:
            line["host_to_host"] = h2h_multi
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
