"""CPU, world_size 2 / 4 / 8 over gloo: the N>1 paths of pipeline.run_clip -- every batch's sequences shared
out over the ranks in lockstep (a rank may sit a batch out and catches its spectral-norm state up), pano
tile-per-rank with the all-to-all re-shard (shares of ZERO frame-pairs included), the streamed per-batch
variable-length gather (stage functions replaced by CPU stand-ins; index logic is the product's).  The same
cases through ``dist.ThreadWorld`` (N ranks on N threads, the single-GPU emulation of the GPU tests)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_stage2(fps):
    """CPU stand-in for LDATI: per frame-pair k records (k = 1 + pair % 3) = (global pair index,
    checksum of its voxels over the FULL width)."""
    from v2ce_toolbox_amd import glue
    from v2ce_toolbox_amd.LDATI import EVENT_DTYPE

    def begin(vox, first_pair):
        recs = []
        for i in range(vox.shape[0]):
            g = first_pair + i
            r = np.zeros(1 + g % 3, EVENT_DTYPE)
            r["timestamp"] = glue.frame_offset_us(g, fps)
            r["x"] = g % 30000
            r["y"] = int(float(vox[i].double().sum()) * 10) % 30000
            r["polarity"] = vox.shape[-1] % 100
            recs.append(r)
        if not recs:
            return torch.empty(0, dtype=torch.uint8)
        return torch.from_numpy(np.frombuffer(np.concatenate(recs).tobytes(), np.uint8).copy())

    def finish(handle):
        return handle, None
    return begin, finish


def _single(n_frames, infer_type, bs, wf):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_product_glue import FakeModel
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    frames = synth.synthetic_frames(n_frames, 8, wf, seed=3)
    return cli.run(frames, FakeModel(), infer_type=infer_type, width=12, height=8, batch_size=bs, device="cpu",
                   stage2=fake_stage2(30))


def _worker(rank, world, port, n_frames, infer_type, bs, wf, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import v2ce as cli
        from v2ce_toolbox_amd import synth
        frames = synth.synthetic_frames(n_frames, 8, wf, seed=3)
        out = cli.run(frames, FakeModel(), infer_type=infer_type, width=12, height=8, batch_size=bs,
                      device="cpu", stage2=fake_stage2(30))
        # also exercise the raw gather with ragged (and empty) payloads
        payload = torch.full((rank * 5,), rank + 1, dtype=torch.uint8)
        vd = __import__("v2ce_toolbox_amd.dist", fromlist=["x"])
        g = vd.gather_events(payload, dst=0)
        # the split (begin / finish) form bench.py pipelines: two gathers in flight, finished in order
        h1, h2 = vd.EventGather(payload, dst=0), vd.EventGather(payload + 1, dst=0)
        g1, g2 = h1.finish(), h2.finish()
        assert (out is None) == (rank != 0) and (g1 is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(torch.cat(g1), g) and torch.equal(torch.cat(g2), g + 1)
            q.put((out.tobytes(), g.numpy().tolist()))
    finally:
        dist.destroy_process_group()


CASES = [
    # world, frames, infer_type, batch, frame width
    (2, 53, "center", 1, 20),      # batch sharding, overlapped last sequence
    (2, 53, "pano", 2, 20),        # pano, 2 tiles (12 + partial 8): one tile per rank
    (2, 49, "center", 2, 20),
    (4, 53, "pano", 2, 40),        # 4 tiles (last partial: 40 = 3*12 + 4): BASELINE config 4's split
    (4, 85, "pano", 1, 20),        # 2 tiles on 4 ranks: two groups of two, batches sharded over the groups
    (2, 53, "pano", 1, 40),        # 4 tiles on 2 ranks: not a multiple -> batch sharding, tiles serial
    (4, 37, "center", 1, 20),      # more ranks than batches for some (3 batches on 4 ranks)
    (8, 277, "center", 2, 20),     # the scaling bench's largest world: 9 batches of 2 sequences on 8 ranks
    (8, 149, "pano", 1, 40),       # 4 tiles on 8 ranks: two tile groups of four, one sequence per batch: a group idles
    (4, 50, "pano", 1, 40),        # 49 pairs: the overlapped last sequence keeps ONE pair < 4 tiles: empty LDATI shares
    (8, 51, "pano", 2, 40),        # two tile groups; last batch keeps 2 pairs
    (8, 277, "center", 16, 20),    # BASELINE config 3's shape: a big batch shared out over 8 ranks (16 + 2 sequences)
    (2, 17, "center", 4, 20),      # one sequence, two ranks
]


@pytest.mark.parametrize("world,n_frames,infer_type,bs,wf", CASES)
def test_world_equals_single_process(world, n_frames, infer_type, bs, wf):
    single = _single(n_frames, infer_type, bs, wf)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sock:                 # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, infer_type, bs, wf, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == single.tobytes()
    pairs = np.frombuffer(got, single.dtype)["x"]
    assert sorted(set(pairs.tolist())) == list(range(n_frames - 1))       # every frame-pair, in order
    assert np.all(np.diff(pairs) >= 0)
    assert gathered == sum(([r + 1] * (5 * r) for r in range(world)), [])


@pytest.mark.parametrize("world,n_frames,infer_type,bs,wf", CASES)
def test_thread_world_equals_single_process(world, n_frames, infer_type, bs, wf):
    """dist.ThreadWorld (N ranks on N threads of this process) through the same driver."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_product_glue import FakeModel
    from v2ce_toolbox_amd import dist as vd
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    single = _single(n_frames, infer_type, bs, wf)
    frames = synth.synthetic_frames(n_frames, 8, wf, seed=3)
    models = [FakeModel() for _ in range(world)]
    outs = vd.ThreadWorld(world).run(
        lambda comm: cli.run(frames, models[comm.rank], infer_type=infer_type, width=12, height=8, batch_size=bs,
                             device="cpu", stage2=fake_stage2(30), comm=comm))
    assert all(o is None for o in outs[1:])
    assert outs[0].tobytes() == single.tobytes()
    ref = FakeModel()
    _ = cli.run(frames, ref, infer_type=infer_type, width=12, height=8, batch_size=bs, device="cpu", stage2=fake_stage2(30))
    assert all(m.calls == ref.calls for m in models)        # every rank's model ends where the single run's does


class FakeGuardModel:
    """FakeModel + the range-guard interface of V2ce3d (glue.run_guarded): the 'split' arithmetic records the largest input it
    saw as its guard bound; the 'exact' arithmetic adds 1000 to the output, so a rerun is visible in the result."""
    RANGE_GUARD_LIMIT = 3.0

    def __init__(self):
        self.calls, self.precision, self.guard, self.worst, self.reruns = 0, "f16x2", "call", 0.0, 0

    def advance_spectral_norm(self):
        self.calls += 1

    def parameters(self):
        yield torch.zeros(1)

    def __call__(self, x):
        B, L, _, H, W = x.shape
        base = x.mean(dim=2, keepdim=True) + 0.01 * self.calls
        ch = torch.arange(20, dtype=torch.float32).view(1, 1, 20, 1, 1)
        self.calls += 1
        if self.precision == "f16x2":
            self.worst = max(self.worst, float(x.abs().max()))
            return (base * (1 + ch)).contiguous()
        return (base * (1 + ch) + 1000.0).contiguous()

    def range_guard_value(self, reset=True):
        v = self.worst
        if reset:
            self.worst = 0.0
        return v

    def sn_snapshot(self):
        return self.calls

    def sn_restore(self, snap):
        self.calls = snap

    def exact_f32(self):
        m = self

        class Ctx:
            def __enter__(self):
                m.precision, m.reruns = "f32", m.reruns + 1

            def __exit__(self, *exc):
                m.precision = "f16x2"
                return False
        return Ctx()


def _guard_frames():
    from v2ce_toolbox_amd import synth
    frames = (synth.synthetic_frames(70, 8, 20, seed=5) // 4).astype(np.uint8)      # dark: normalised values below 1.5
    frames[60, 3, 7] = 255                                                          # one bright pixel in the LAST sequences only
    return frames


def _guard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from v2ce_toolbox_amd import v2ce as cli
        m = FakeGuardModel()
        out = cli.run(_guard_frames(), m, infer_type="center", width=12, height=8, batch_size=4, device="cpu", stage2=fake_stage2(30))
        q.put((rank, m.reruns, m.calls, None if out is None else out.tobytes()))
    finally:
        dist.destroy_process_group()


def test_range_guard_decision_is_collective():
    """glue.run_guarded under N ranks: the guard bound is reduced (max) over the ranks, so when ONE rank's
    share trips it EVERY rank repeats the clip on the exact arithmetic (the clip contains collectives: a split decision would
    hang) and rank 0 gets what the single-process run gets.  gloo world 2 and ThreadWorld 2; only the rank that holds the last
    sequences sees the bright pixel."""
    from v2ce_toolbox_amd import dist as vd
    from v2ce_toolbox_amd import v2ce as cli
    kw = dict(infer_type="center", width=12, height=8, batch_size=4, device="cpu", stage2=fake_stage2(30))
    ref = FakeGuardModel()
    single = cli.run(_guard_frames(), ref, **kw)
    assert ref.reruns == 1 and single["y"].max() > 0
    quiet = FakeGuardModel()
    cli.run((_guard_frames() // 255).astype(np.uint8), quiet, **kw)
    assert quiet.reruns == 0
    # threads
    models = [FakeGuardModel() for _ in range(2)]
    outs = vd.ThreadWorld(2).run(lambda comm: cli.run(_guard_frames(), models[comm.rank], comm=comm, **kw))
    assert [m.reruns for m in models] == [1, 1] and all(m.calls == ref.calls for m in models)
    assert outs[1] is None and outs[0].tobytes() == single.tobytes()
    # processes over gloo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [1, 1] and [g[2] for g in got] == [ref.calls] * 2
    assert got[0][3] == single.tobytes() and got[1][3] is None


# ------------------------------------------------------------------------------------------------
# round 4: gather modes, failure protocol, bounded receive buffers
# ------------------------------------------------------------------------------------------------
def _spawn(target, world, args=(), timeout=180):
    import socket
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(got)


def _mode_worker(rank, world, port, q, mode, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), V2CE_GATHER=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import synth
        from v2ce_toolbox_amd import v2ce as cli
        frames = synth.synthetic_frames(85, 8, 20, seed=3)
        kw = dict(infer_type="center", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
        out = cli.run(frames, FakeModel(), **kw)
        path = os.path.join(out_dir, f"{mode}.npz")
        n = cli.run(frames, FakeModel(), out_path=path, **kw)         # the streamed .npz: every rank writes its own slice
        q.put((rank, None if out is None else bytes(out.tobytes()), n))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_gather_modes_equal_single_process(mode, tmp_path):
    """Both ways of bringing the ranks' records together -- 'host' (every rank writes its own slice of the shared host
    segment / of the streamed .npz; only byte counts are exchanged) and 'device' (gather on rank 0, rank 0 downloads) --
    give the single-process bytes, as an array and as a file."""
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_product_glue import FakeModel
    frames = synth.synthetic_frames(85, 8, 20, seed=3)
    single = cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
    got = _spawn(_mode_worker, 3, (mode, str(tmp_path)))
    assert got[0][1] == single.tobytes() and got[1][1] is None and got[2][1] is None
    assert got[0][2] == len(single)
    z = np.load(tmp_path / f"{mode}.npz")["event_stream"]
    assert z.tobytes() == single.tobytes() and not os.path.exists(tmp_path / f"{mode}.npz.part")


def _window_worker(rank, world, port, q, window):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), V2CE_GATHER="host")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import pipeline, synth
        from v2ce_toolbox_amd import v2ce as cli
        seen = []
        orig = pipeline.vdist.HostDirectGather

        class Spy(orig):                                         # (how many bytes went through the window on this rank)
            def finalize(self):
                seen.append((self.reg_bytes, self.dma_bytes))
                return super().finalize()
        pipeline.vdist.HostDirectGather = Spy
        pipeline._registered_window_bytes = lambda n_pairs, device: window
        frames = synth.synthetic_frames(85, 8, 20, seed=3)
        out = cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
        q.put((rank, None if out is None else bytes(out.tobytes()), seen[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("window", ["none", "part", "straddle", "all"])
def test_host_gather_window_covers_part_of_the_clip(window):
    """gather='host' with the shared window (the mapping every rank fills directly: page-locked and written by DMA on GPU ranks,
    a plain mapping here) covering nothing, the first part, a size that cuts a rank's piece, or all of the clip: the pieces that
    end inside the window go there, the rest takes the pwrite path behind it, and rank 0's array is the single-process bytes in
    every case (the three window cases had run on one GPU only)."""
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_product_glue import FakeModel
    frames = synth.synthetic_frames(85, 8, 20, seed=3)
    single = cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
    total = len(single.tobytes())
    size = {"none": 0, "part": (total // 3) & ~63, "straddle": total // 2 + 7, "all": total + 4096}[window]
    got = _spawn(_window_worker, 2, (size,))
    assert got[0][1] == single.tobytes() and got[1][1] is None
    through = sum(g[2][1] for g in got)
    assert all(g[2][0] == size for g in got)
    if window == "none":
        assert through == 0
    elif window == "all":
        assert through == total
    else:
        assert 0 < through < total and through <= size


def failing_stage2(fps, bad_rank, bad_pair):
    begin, finish = fake_stage2(fps)

    def begin2(vox, first_pair):
        return (begin(vox, first_pair), first_pair, vox.shape[0])

    def finish2(handle):
        packed, first_pair, n = handle
        if dist.get_rank() == bad_rank and first_pair <= bad_pair < first_pair + n:
            raise ValueError("stage 2 failed on purpose")
        return packed, None
    return begin2, finish2


def _failure_worker(rank, world, port, q, mode, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), V2CE_GATHER=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import dist as vd
        from v2ce_toolbox_amd import synth
        from v2ce_toolbox_amd import v2ce as cli
        frames = synth.synthetic_frames(101, 8, 20, seed=3)            # 7 sequences, batches of 2: rank 1 holds pairs 48..63 in batch 1
        path = os.path.join(out_dir, "f.npz")
        try:
            cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu",
                    stage2=failing_stage2(30, 1, 50), out_path=path)
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "own:" + str(e)))
        except vd.RankFailure as e:
            q.put((rank, "peer:" + str(e.ranks)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_rank_failure_stops_every_rank(mode, tmp_path):
    """a stage-2 error on ONE rank in the middle of a clip.  The rank reports it with its next byte count
    (-1), every rank stops at the same step -- the failed rank with its own error, the others with RankFailure naming it --,
    nobody waits for a collective the failed rank never joins, and no file (complete or partial) is left behind."""
    got = _spawn(_failure_worker, 2, (mode, str(tmp_path)), timeout=120)
    assert got[0] == (0, "peer:[1]") and got[1] == (1, "own:stage 2 failed on purpose")
    assert os.listdir(tmp_path) == []


def _pool_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from v2ce_toolbox_amd import dist as vd
        seen, caps = [], []
        g = vd.StreamedGather(0, None, (lambda pieces, stream: seen.append([int(p.sum()) for p in pieces])) if rank == 0 else None)
        sizes = [24 << 20, 30 << 20, 28 << 20, 30 << 20, 1 << 10, 29 << 20]       # "pano-sized" steps, scaled to the CPU box
        for k, n in enumerate(sizes):
            g.submit(torch.full((n + rank,), (k + rank) % 7, dtype=torch.uint8))
            caps.append((g.pool.get("cap", 0), id(g.pool.get("recv"))))
        g.drain()
        ok = True
        if rank == 0:
            ok = seen == [[(k % 7) * sizes[k], ((k + 1) % 7) * (sizes[k] + 1)] for k in range(len(sizes))]
        # one allocation for the first step, one growth for the 30 MiB step, then reuse: the receive buffers do not scale
        # with the number of steps in flight
        q.put((rank, ok, len({c for c in caps if c[0]})))
    finally:
        dist.destroy_process_group()


def test_device_gather_reuses_bounded_receive_buffers():
    """the padded gather's send / receive buffers live in a pool sized from the largest step so far
    (+25 %), not in fresh allocations per step: six steps of up to 30 MiB per rank allocate twice."""
    got = _spawn(_pool_worker, 2)
    assert all(ok for _, ok, _ in got) and all(n <= 2 for _, _, n in got), got


def test_single_rank_failure_leaves_no_partial_file(tmp_path):
    """a clip that dies half-way through a streamed .npz must not leave a truncated archive (nor its
    .part) behind, and the writer thread must be gone -- single process, error raised by stage 2 in the third batch."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_product_glue import FakeModel
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    begin, finish = fake_stage2(30)
    calls = {"n": 0}

    def finish_bad(handle):
        calls["n"] += 1
        if calls["n"] == 3:
            raise ValueError("stage 2 failed on purpose")
        return finish(handle)
    frames = synth.synthetic_frames(101, 8, 20, seed=3)
    before = threading.active_count()
    with pytest.raises(ValueError, match="on purpose"):
        cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu",
                stage2=(begin, finish_bad), out_path=str(tmp_path / "f.npz"))
    assert os.listdir(tmp_path) == [] and threading.active_count() <= before
    # and the same call without the fault writes the file under its final name only
    n = cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu",
                stage2=(begin, finish), out_path=str(tmp_path / "f.npz"))
    assert os.listdir(tmp_path) == ["f.npz"] and len(np.load(tmp_path / "f.npz")["event_stream"]) == n


# ------------------------------------------------------------------------------------------------
# round 5: failures that used to leave the peers inside a collective
# ------------------------------------------------------------------------------------------------
def _pano_failure_worker(rank, world, port, q, bad_rank, bad_call):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import dist as vd
        from v2ce_toolbox_amd import synth
        from v2ce_toolbox_amd import v2ce as cli

        class FailingModel(FakeModel):
            def __call__(self, x):
                self.n_real = getattr(self, "n_real", 0) + 1
                if rank == bad_rank and self.n_real == bad_call:
                    raise ValueError("model failed on purpose")
                return super().__call__(x)
        frames = synth.synthetic_frames(101, 8, 40, seed=3)           # 4 tiles on 4 ranks: one tile per rank, 4 batches of 2 sequences
        try:
            cli.run(frames, FailingModel(), infer_type="pano", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "own:" + str(e)))
        except vd.RankFailure as e:
            q.put((rank, "peer:" + str(e.ranks)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank,bad_call", [(2, 2), (0, 1), (3, 4)])
def test_pano_tile_rank_failure_stops_every_rank(bad_rank, bad_call):
    """A tile rank whose MODEL call raises, i.e. between the start of a batch and the group's all_to_all: it keeps taking
    part in the tile exchanges (zeros of the right shape), reports the failure with its next byte count, and every rank
    stops at the same step -- nobody is left inside an unmatched collective (the old protocol hung here until the watchdog)."""
    got = _spawn(_pano_failure_worker, 4, (bad_rank, bad_call), timeout=150)
    for r, msg in got:
        assert msg == ("own:model failed on purpose" if r == bad_rank else f"peer:[{bad_rank}]"), got


def _io_failure_worker(rank, world, port, q, mode, out_dir, what):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), V2CE_GATHER=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        import errno
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import dist as vd
        from v2ce_toolbox_amd import pipeline, synth
        from v2ce_toolbox_amd import v2ce as cli
        calls = {"n": 0}
        if what == "pwrite" and rank == 1:                     # rank 1's writer thread: the segment is full at its second piece
            real = os.pwrite

            def full(fd, data, off):
                calls["n"] += 1
                if calls["n"] >= 2:
                    raise OSError(errno.ENOSPC, "No space left on device (on purpose)")
                return real(fd, data, off)
            vd.os.pwrite = full
        if what == "sink" and rank == 0:                       # rank 0's sink (device gather): fails at the third batch
            real_push = pipeline.EventSink.push

            def push(self, packed, n_pairs, keep=(), src_stream=None):
                calls["n"] += 1
                if calls["n"] >= 5:
                    raise OSError(errno.EIO, "sink failed on purpose")
                return real_push(self, packed, n_pairs, keep, src_stream=src_stream)
            pipeline.EventSink.push = push
        if what == "open" and rank == 1:                       # rank 1 cannot open the shared segment
            real_open = os.open

            def no_open(path, flags, *a):
                if "v2ce" in str(path) or str(path).endswith(".part"):
                    raise OSError(errno.EACCES, "cannot open the shared segment (on purpose)")
                return real_open(path, flags, *a)
            vd.os.open = no_open
        frames = synth.synthetic_frames(101, 8, 20, seed=3)
        try:
            cli.run(frames, FakeModel(), infer_type="center", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
            q.put((rank, "no error"))
        except OSError as e:
            q.put((rank, "own:" + str(e.strerror)))
        except vd.RankFailure as e:
            q.put((rank, "peer:" + str(e.ranks)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,what,bad", [("host", "pwrite", 1), ("host", "open", 1), ("device", "sink", 0)])
def test_local_io_failure_stops_every_rank(mode, what, bad, tmp_path):
    """A rank-local I/O error -- the writer thread of gather='host' (ENOSPC), opening the shared segment, rank 0's sink under
    gather='device' -- is reported through the failure flag of the next exchange: the failing rank raises its own OSError,
    every other rank RankFailure naming it, at the same step; none is left waiting in a collective."""
    got = _spawn(_io_failure_worker, 2, (mode, str(tmp_path), what), timeout=150)
    for r, msg in got:
        assert msg.startswith("own:") if r == bad else msg == f"peer:[{bad}]", got
