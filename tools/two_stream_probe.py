"""Probe (round 6): would two half-batches of a forward call on two streams fill each other's launch tails?
Two V2ce3d replicas (own state: range slots, spectral-norm vectors), 2 sequences each, on two streams -- against one replica with
4 sequences on one stream.  Timing only (the replicas' spectral-norm states advance independently; not the product's schedule)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from v2ce_toolbox_amd import synth  # noqa: E402
from v2ce_toolbox_amd.v2ce_3d import V2ce3d  # noqa: E402


def model():
    m = V2ce3d()
    m.load_state_dict(synth.make_state_dict(0), strict=True)
    return m.eval().to("cuda")


H, W, L = 260, 346, 16
x = torch.randn(4, L, 2, H, W, device="cuda") * 0.5
m1 = model()
ma, mb = model(), model()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x[:2].contiguous(), x[2:].contiguous()


def one(n):
    for _ in range(n):
        m1(x)


def two(n):
    for _ in range(n):
        with torch.cuda.stream(sa):
            ma(xa)
        with torch.cuda.stream(sb):
            mb(xb)


for name, fn in (("one stream, 4 sequences", one), ("two streams, 2 + 2 sequences", two), ("one stream, 4 sequences", one), ("two streams, 2 + 2 sequences", two)):
    fn(3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(20)
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t) / 20:.3f} ms per 64 frame-pairs")
