"""CPU, world_size 2 over gloo: the N>1 path -- variable-length event gather and the sequence
sharding of v2ce.run (stage functions replaced by CPU stand-ins; index logic is the product's)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_events_from_voxels(pred_voxel, fps, stage2_batch_size=24, seed=0, rng="philox", first_pair=0):
    """CPU stand-in for LDATI: one 13-byte record per frame-pair = (global pair index, checksum)."""
    from v2ce_toolbox_amd import glue
    from v2ce_toolbox_amd.LDATI import EVENT_DTYPE
    L = pred_voxel.shape[0]
    rec = np.zeros(L, EVENT_DTYPE)
    for i in range(L):
        rec["timestamp"][i] = glue.frame_offset_us(first_pair + i, fps)
        rec["x"][i] = (first_pair + i) % 30000
        rec["y"][i] = int(float(pred_voxel[i].double().sum()) * 10) % 30000
    return [torch.from_numpy(np.frombuffer(rec.tobytes(), np.uint8).copy())], np.ones(L, np.int64)


def _worker(rank, world, port, n_frames, infer_type, bs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_product_glue import FakeModel
        from v2ce_toolbox_amd import v2ce as cli
        from v2ce_toolbox_amd import synth
        cli.events_from_voxels = _fake_events_from_voxels
        frames = synth.synthetic_frames(n_frames, 8, 20, seed=3)
        out = cli.run(frames, FakeModel(), infer_type=infer_type, width=12, height=8, batch_size=bs,
                      device="cpu")
        # also exercise the raw gather with ragged (and empty) payloads
        payload = torch.full((rank * 5,), rank + 1, dtype=torch.uint8)
        g = __import__("v2ce_toolbox_amd.dist", fromlist=["x"]).gather_events(payload, dst=0)
        if rank == 0:
            q.put((out.tobytes(), g.numpy().tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,infer_type,bs", [(53, "center", 1), (53, "pano", 2), (49, "center", 2)])
def test_world2_equals_single_process(n_frames, infer_type, bs):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_product_glue import FakeModel
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    frames = synth.synthetic_frames(n_frames, 8, 20, seed=3)
    orig = cli.events_from_voxels
    cli.events_from_voxels = _fake_events_from_voxels
    try:
        single = cli.run(frames, FakeModel(), infer_type=infer_type, width=12, height=8, batch_size=bs,
                         device="cpu")
    finally:
        cli.events_from_voxels = orig
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_frames
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, infer_type, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == single.tobytes()
    assert len(single) == n_frames - 1
    assert gathered == [2] * 5
