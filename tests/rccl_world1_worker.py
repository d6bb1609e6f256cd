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
    dist.destroy_process_group()
    print("rccl world-of-one ok")


if __name__ == "__main__":
    main()
