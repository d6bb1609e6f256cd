#!/usr/bin/env python3
"""Time one conv layer with explicit output boxes (tools/conv_bench.py's `tile` override).
    PRECISION=f16x2 FUSE=sc python tools/tile_probe.py dec3.conv1 4,4,32 4,8,16 8,4,16 2,8,32"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.conv_bench as cb   # noqa: E402
name = sys.argv[1]
for spec in sys.argv[2:]:
    t = tuple(int(v) for v in spec.split(","))
    try:
        print(spec, cb.run(name, tile=t, iters=8))
    except Exception as e:   # box does not fit the kernel's LDS plane
        print(spec, "not possible:", str(e)[:80])
