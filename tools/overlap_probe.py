import sys, time, torch
sys.path.insert(0, '.')
import bench
from v2ce_toolbox_amd import synth
dev = torch.device("cuda:0")
m = bench.fresh_model("f16x2", dev)
x = bench.make_inputs(4, 0, dev)
big = torch.empty(152 << 20, dtype=torch.uint8, device=dev)
host = torch.empty(152 << 20, dtype=torch.uint8, pin_memory=True)
cs = torch.cuda.Stream()
def run(copy, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        m(x)
        if copy:
            with torch.cuda.stream(cs):
                host.copy_(big, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(False, 3); run(True, 3)
print("model only ms/step", run(False)); print("model + concurrent 152 MB D2H ms/step", run(True))
torch.cuda.synchronize(); t0=time.perf_counter(); host.copy_(big, non_blocking=True); torch.cuda.synchronize(); print("D2H alone ms", (time.perf_counter()-t0)*1e3)
