"""GPU: BASELINE configs C3 and C4 at FULL size with the real kernels on the one GPU a box has, and every
product RCCL call under a world of one.

* C3 -- 2048 synthetic 346x260 frames, ``-b 32`` (v2ce.py:149-154,211-239,365): the single-rank run of the
  product driver (sequence plan, overlapped last sequence, per-frame offsets, global time order), then the
  SAME driver as 8 ranks -- ``dist.ThreadWorld``: 8 threads of this process, each with its own model replica,
  its share of every batch's sequences (4 of 32), its spectral-norm state, the streamed gather to rank 0 --
  must reproduce the single run byte for byte.
* C4 -- 129 frames 1384x260, ``-t pano -b 8`` (v2ce.py:100-129): single rank (four tiles serially per batch,
  call index 4k+g) vs 4 tile-ranks (one W-tile per rank, all-to-all W-tiles -> frame-pairs, full-width
  LDATI): byte-equal; one sequence's four tiles against oracle/unet.py at 1e-5.
* RCCL -- ``torchrun --nproc-per-node 1`` with V2CE_FORCE_DIST=1: the CLI's distributed path (StreamedGather:
  all_gather_into_tensor + gather on the communication stream, all_reduce of the range guard) and a direct
  worker for the list-form all_to_all of ``tiles_to_pairs`` on ``nccl``.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import glue as OG
from oracle import unet as U
from v2ce_toolbox_amd import dist as vdist
from v2ce_toolbox_amd import glue, synth
from v2ce_toolbox_amd import v2ce as cli

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 260, 346
TOL = 1e-5


def fresh_model():
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d()
    m.load_state_dict(synth.make_state_dict(0), strict=True)
    return m.eval().to("cuda")


def same_bytes(a: np.ndarray, b: np.ndarray) -> bool:
    """Byte equality of two big structured arrays without tobytes() copies."""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    ua, ub = a.view(np.uint8).reshape(-1), b.view(np.uint8).reshape(-1)
    step = 1 << 28
    return all(np.array_equal(ua[i:i + step], ub[i:i + step]) for i in range(0, len(ua), step))


def run_world(world, frames, **kw):
    """The product driver (v2ce.run -> glue.run_guarded -> pipeline.run_clip) as `world` ranks on threads."""
    models = [fresh_model() for _ in range(world)]
    outs = vdist.ThreadWorld(world).run(lambda comm: cli.run(frames, models[comm.rank], comm=comm, **kw), device="cuda")
    assert all(o is None for o in outs[1:])
    return outs[0], models


def test_c3_2048_frames_batch32_single_and_eight_ranks():
    n = 2048
    frames = synth.synthetic_frames(n, H, W)
    num, mode, starts = glue.sequence_plan(n)
    assert (num, mode, int(starts[-1])) == (128, 15, 2031)           # v2ce.py:149-154: last sequence starts at 2031
    kw = dict(infer_type="center", batch_size=32, fps=30, seed=11)
    single_model = fresh_model()
    single = cli.run(frames, single_model, **kw).copy()             # (the driver may hand out a reused pinned buffer)
    assert single_model.calls == 4                                   # 128 sequences / 32: four model calls
    ts = single["timestamp"]
    assert np.all(np.diff(ts) >= 0)                                  # frame-major, bins ascending, sorted segments
    # frame-pair i occupies [int(i/fps*1e6), int((i+1)/fps*1e6)] (v2ce.py:365 offset + LDATI's f32 times in [0, 1/fps]: the
    # last microsecond of a frame may round onto the next frame's offset, never beyond it)
    off = np.array([glue.frame_offset_us(i, 30) for i in range(n)], np.int64)
    assert off[123] == int(123 * 1 / 30 * 1e6) and off[2046] == int(2046 * 1 / 30 * 1e6)
    assert ts[0] >= 0 and ts[-1] <= off[n - 1] + 2
    per_pair = np.diff(np.searchsorted(ts, off, side="left"))
    assert len(per_pair) == n - 1 and per_pair.min() > 1000 and 0 <= len(single) - per_pair.sum() < 64
    for i in (123, 2046):
        seg = single[np.searchsorted(ts, off[i]):np.searchsorted(ts, off[i + 1])]
        assert len(seg) == per_pair[i] > 1000 and seg["timestamp"].min() >= off[i]
        assert 0 <= seg["x"].min() and seg["x"].max() < W and 0 <= seg["y"].min() and seg["y"].max() < H
    got, models = run_world(8, frames, **kw)
    assert all(m.calls == 4 for m in models)
    assert same_bytes(got, single)


def test_c3_plan_shares():
    """What each of the 8 ranks runs per batch in C3: four sequences; the overlapped pairs drop on rank 7 only."""
    from v2ce_toolbox_amd import pipeline
    plans = pipeline.plan_batches(2048, 16, 32)
    assert [len(p.seqs) for p in plans] == [32] * 4 and plans[-1].drop == 1
    for bp in plans:
        shares = [pipeline.shard_of_batch(bp, 16, r, 8) for r in range(8)]
        assert [len(s.seqs) for s in shares] == [4] * 8
        assert [s.first_pair for s in shares] == [bp.first_pair + 64 * r for r in range(8)]
        assert sum(s.n_pairs for s in shares) == bp.n_pairs and [s.drop for s in shares[:-1]] == [0] * 7


def test_c4_pano_1384_batch8_single_and_four_tile_ranks():
    n, WF = 129, 1384
    frames = synth.synthetic_frames(n, H, WF, seed=21)
    kw = dict(infer_type="pano", batch_size=8, fps=30, seed=12)
    single_model = fresh_model()
    single = cli.run(frames, single_model, **kw).copy()
    assert single_model.calls == 4                                   # one batch of 8 sequences, four tile calls
    ts = single["timestamp"]
    assert np.all(np.diff(ts) >= 0) and single["x"].max() == WF - 1 and single["y"].max() == H - 1
    got, models = run_world(4, frames, **kw)
    assert all(m.calls == 4 for m in models)
    assert same_bytes(got, single)
    # stage 1 of one sequence, tile by tile, against the oracle: tile g of batch 0 is the reference's call g
    # (v2ce.py:103-126: one model call per tile; the spectral-norm state advances between them)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    x = OG.preprocess(frames[:17])[None]                             # [1,16,2,H,1384]
    sd = U.clone_state(synth.make_state_dict(0))
    m = fresh_model()
    for g in range(4):
        xt = np.ascontiguousarray(x[..., g * W:(g + 1) * W])
        want = U.forward(sd, torch.from_numpy(xt)).contiguous().numpy()   # advances sd's u / v like the reference
        got_t = m(torch.from_numpy(xt).cuda()).cpu().numpy()
        err = np.abs(got_t - want) - TOL * np.abs(want)
        assert err.max() <= TOL, (g, float(err.max()))


def free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def torchrun(args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


@pytest.mark.parametrize("infer_type,wf", [("center", 48), ("pano", 112)])
def test_cli_under_torchrun_world_of_one_nccl(tmp_path, infer_type, wf):
    """v2ce.py with V2CE_FORCE_DIST=1 under torchrun: init_process_group('nccl'), dist.StreamedGather
    (all_gather_into_tensor of the byte counts + padded gather on the communication stream, rank 0 appending into the
    pinned sink), the all_reduce behind run_guarded -- same file as the plain single-process CLI."""
    outs = []
    np.save(tmp_path / "f.npy", synth.synthetic_frames(70, 32, wf, seed=9))
    for forced in (False, "device", "host"):               # both gather modes of the distributed path (RCCL world of one)
        out = tmp_path / (forced or "s")
        args = [os.path.join(ROOT, "v2ce.py"), "--npy_frames", str(tmp_path / "f.npy"), "--height", "32", "--width", "48",
                "--synthetic_weights", "0", "-o", str(out), "-b", "2", "--seed", "3", "-t", infer_type,
                "--write_event_frame_video", "false"]
        if forced:
            torchrun(args, {"V2CE_FORCE_DIST": "1", "V2CE_GATHER": forced})
        else:
            r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=900, cwd=ROOT)
            assert r.returncode == 0, r.stderr[-2000:]
        files = [f for f in os.listdir(out) if f.endswith("-events.npz")]
        assert len(files) == 1
        outs.append(np.load(out / files[0])["event_stream"])
    assert len(outs[0]) > 1000 and outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes()


def test_rccl_collectives_world_of_one():
    """tests/rccl_world1_worker.py under torchrun: tiles_to_pairs (list-form dist.all_to_all inside a new_group),
    StreamedGather with ragged and empty payloads, TorchComm.max_float -- all on the nccl backend."""
    r = torchrun([os.path.join(ROOT, "tests", "rccl_world1_worker.py")], {})
    assert "rccl world-of-one ok" in r.stdout



def test_bench_times_both_exchanges_under_torchrun():
    """bench.py's multi-rank path on nccl at a world of one (V2CE_BENCH_FORCE_DIST=1) with V2CE_BENCH_BOTH_GATHERS=1 -- what
    the driver's first 8-GPU run takes by itself (round 6): the JSON line carries `value` for the default exchange ('device':
    the RCCL gather north_star names) and, under gather.other_mode, the same steps timed with 'host'."""
    import json
    r = torchrun([os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-exact-f32",
                  "--no-host-to-host"], {"V2CE_BENCH_FORCE_DIST": "1", "V2CE_BENCH_BOTH_GATHERS": "1"})
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    g = line["gather"]
    assert g["mode"] == "device" and g["inside_timed_region"] and line["value"] > 0 and line["rccl_world"] == 1
    other = g["other_mode"]
    assert other["mode"] == "host" and "error" not in other and other["value"] > 0 and other["gathered_bytes_per_step"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the real multi-process RCCL path (ADVICE r3)")
@pytest.mark.parametrize("mode", ["host", "device"])
def test_two_ranks_nccl_equal_single_process(tmp_path, mode):
    """Two processes, two GPUs, RCCL: the CLI under torchrun against the single-process CLI, byte for byte, in both gather
    modes (communication-stream ordering, buffer lifetimes and the concurrent use of the world and tile communicators are
    only exercised here -- the ThreadWorld emulation shares one stream).  Skipped on the 1-GPU boxes of this pool."""
    np.save(tmp_path / "f.npy", synth.synthetic_frames(150, 64, 160, seed=9))
    outs = []
    for nproc in (1, 2):
        out = tmp_path / f"o{nproc}"
        args = [os.path.join(ROOT, "v2ce.py"), "--npy_frames", str(tmp_path / "f.npy"), "--height", "64", "--width", "80",
                "--synthetic_weights", "0", "-o", str(out), "-b", "4", "--seed", "3", "-t", "pano", "--write_event_frame_video", "false"]
        env = dict(os.environ, V2CE_GATHER=mode)
        env.pop("MASTER_PORT", None)
        cmd = ([sys.executable] if nproc == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(free_port())]) + args
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        files = [f for f in os.listdir(out) if f.endswith("-events.npz")]
        outs.append(np.load(out / files[0])["event_stream"])
    assert len(outs[0]) > 1000 and outs[0].tobytes() == outs[1].tobytes()
