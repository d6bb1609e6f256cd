"""Worker of tests/test_gpu_multirank.py::test_rccl_collectives_world_of_one (run under torchrun, one rank, one GPU):
the product's collectives on the nccl (= RCCL) backend with a world of one."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from v2ce_toolbox_amd import dist as vd          # noqa: E402


def main():
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group("nccl")
    comm = vd.default_comm(force=True)
    assert isinstance(comm, vd.TorchComm) and (comm.rank, comm.world) == (0, 1)
    # tiles_to_pairs: a one-tile group; the list-form all_to_all of RCCL inside a new_group
    grp = comm.tile_group(1)
    part = torch.arange(5 * 2 * 10 * 4 * 6, dtype=torch.float32, device="cuda").reshape(5, 2, 10, 4, 6)
    vox, lo = comm.tiles_to_pairs(part, [6], 0, grp)
    assert lo == 0 and torch.equal(vox, part)
    empty, lo = comm.tiles_to_pairs(part[:0], [6], 0, grp)           # a share of zero frame-pairs
    assert empty.shape == (0, 2, 10, 4, 6)
    # StreamedGather: three steps (ragged, empty, ragged), pieces handed over on the communication stream
    seen = []

    def on_pieces(pieces, stream):
        assert stream is not None and len(pieces) == 1
        stream.synchronize()
        seen.append(pieces[0].cpu().clone())
    g = comm.streamed_gather(on_pieces, dst=0)
    payloads = [torch.full((13 * 7,), 3, dtype=torch.uint8, device="cuda"), torch.empty(0, dtype=torch.uint8, device="cuda"),
                torch.arange(26, dtype=torch.uint8, device="cuda")]
    for p in payloads:
        g.submit(p)
    g.drain()
    assert len(seen) == 3 and all(torch.equal(a, b.cpu()) for a, b in zip(seen, payloads)) and g.bytes_last == 26
    assert torch.equal(vd.gather_events(payloads[2], dst=0), payloads[2])
    assert comm.max_float(2.5, device="cuda") == 2.5
    comm.barrier()
    # gather='host' with the REGISTERED shared segment (round 5): every piece goes from the GPU straight into the page-locked
    # mapping of the tmpfs segment at its offset of the clip's record stream; a window that is too small for the clip hands
    # the rest to the staging + pwrite path.  Real kernels, the product driver; bytes equal to the single-process run.
    import numpy as np
    from v2ce_toolbox_amd import pipeline, synth
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d

    def model():
        m = V2ce3d(guard="deferred")
        m.load_state_dict(synth.make_state_dict(0))
        return m.eval().to("cuda")
    frames = synth.synthetic_frames(49, 64, 96, seed=5)
    kw = dict(infer_type="center", width=96, height=64, batch_size=1, fps=30, seed=7, device="cuda")
    single = pipeline.run_clip(frames, model(), comm=vd.LocalComm(), **kw)
    for seg_mb, expect_dma in (("64", "all"), ("4", "some"), ("0", "none")):   # 7.7 MB in three pieces
        os.environ["V2CE_HOST_SEGMENT_MB"] = seg_mb
        seen_dma = []
        real = vd.HostDirectGather.finalize

        def finalize(self, _real=real, _seen=seen_dma):
            _seen.append(self.dma_bytes)
            return _real(self)
        vd.HostDirectGather.finalize = finalize
        try:
            got = pipeline.run_clip(frames, model(), comm=comm, gather="host", **kw)
        finally:
            vd.HostDirectGather.finalize = real
        assert got.tobytes() == single.tobytes(), seg_mb
        total = single.nbytes
        assert len(seen_dma) == 1 and {"all": seen_dma[0] == total, "some": 0 < seen_dma[0] < total, "none": seen_dma[0] == 0}[expect_dma], \
            (seg_mb, seen_dma, total)
        del got
    os.environ.pop("V2CE_HOST_SEGMENT_MB", None)
    assert len(single) > 1000 and np.all(np.diff(single["timestamp"]) >= 0)
    dist.destroy_process_group()
    print("rccl world-of-one ok")


if __name__ == "__main__":
    main()
