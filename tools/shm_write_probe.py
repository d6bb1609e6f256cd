"""Host side of dist.HostDirectGather: how fast pinned staging buffers reach a tmpfs segment.
  pwrite : os.pwrite into one growing file (the kernel serialises writers of one inode: threads / ranks do not add up)
  mmap   : stores through a shared mapping of a pre-sized sparse file (page faults allocate the pages; scales with threads)
and what page-locking a staging buffer costs."""
import concurrent.futures as cf
import mmap
import os
import time

import numpy as np
import torch

n = 151 << 20
t0 = time.perf_counter(); buf = torch.empty(n, dtype=torch.uint8, pin_memory=True); t1 = time.perf_counter()
print(f"pin {n >> 20} MiB: {1e3 * (t1 - t0):.1f} ms")
src = buf.numpy()
path = "/dev/shm/v2ce_probe.bin"
STEPS = 8
for threads in (1, 4):
    pool = cf.ThreadPoolExecutor(threads)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    chunk = n // threads
    t0 = time.perf_counter()
    for k in range(STEPS):
        list(pool.map(lambda i: os.pwrite(fd, memoryview(src[i * chunk:(i + 1) * chunk]), k * n + i * chunk), range(threads)))
    dt = time.perf_counter() - t0
    print(f"pwrite, {threads} thread(s), fresh pages: {STEPS * n / dt / 1e9:.2f} GB/s")
    os.close(fd); os.unlink(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    os.ftruncate(fd, 64 << 30)
    mm = np.frombuffer(mmap.mmap(fd, 64 << 30), np.uint8)
    t0 = time.perf_counter()
    for k in range(STEPS):
        def cp(i, k=k):
            mm[k * n + i * chunk:k * n + (i + 1) * chunk] = src[i * chunk:(i + 1) * chunk]
        list(pool.map(cp, range(threads)))
    dt = time.perf_counter() - t0
    print(f"mmap,   {threads} thread(s), fresh pages: {STEPS * n / dt / 1e9:.2f} GB/s")
    del mm
    os.close(fd); os.unlink(path)
