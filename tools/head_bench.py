#!/usr/bin/env python3
"""Time the UNet head convolution (Cin 2 -> 32, 346x260, B = 4 x T = 16) alone."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.conv_bench as cb   # noqa: E402
cb.LAYERS["head"] = (2, 0, 0, 0, 32, 3, 1)
print(cb.run("head", iters=20))
