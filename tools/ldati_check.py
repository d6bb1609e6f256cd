#!/usr/bin/env python3
"""Debug aid (GPU box): HIP LDATI vs the C oracle over a list of cases, reporting WHERE the first
difference is instead of a bare assert.  `python tools/ldati_check.py [--time]`"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ldati as O                                    # noqa: E402
from v2ce_toolbox_amd import synth                               # noqa: E402
from v2ce_toolbox_amd.LDATI import ldati_device                  # noqa: E402

CASES = [((1, 1, 1), "stress", 30), ((1, 5, 7), "stress", 30), ((3, 12, 14), "sparse", 30), ((2, 33, 47), "stress", 25),
         ((2, 64, 80), "stress", 120), ((4, 5, 129), "frac", 24), ((2, 37, 167), "stress", 30),
         ((1, 260, 346), "sparse", 30), ((2, 260, 346), "stress", 30), ((2, 40, 50), "stress", 10)]


def compare(ev, want, tag):
    seg, ts, x, y, p = want
    ok = True
    if not np.array_equal(ev.seg_counts, seg):
        print(f"  {tag}: seg_counts differ: got {ev.seg_counts.reshape(-1)[:18]} want {seg.reshape(-1)[:18]}")
        return False
    offs = np.concatenate([[0], np.cumsum(seg.reshape(-1))])
    for name, got, exp in (("ts", ev.ts, ts), ("x", ev.x, x), ("y", ev.y, y), ("p", ev.p, p)):
        g = got.cpu().numpy()
        if g.shape != exp.shape:
            print(f"  {tag}: {name} shape {g.shape} vs {exp.shape}")
            ok = False
            continue
        bad = np.nonzero(g != exp)[0]
        if bad.size:
            i = int(bad[0])
            s = int(np.searchsorted(offs, i, side="right") - 1)
            print(f"  {tag}: {name}: {bad.size} of {g.size} differ; first at {i} (segment {s} = frame {s // 9} bin {s % 9}, "
                  f"offset {i - offs[s]} of {offs[s + 1] - offs[s]}): got {g[i:i + 6]} want {exp[i:i + 6]}")
            segs = np.unique(np.searchsorted(offs, bad, side="right") - 1)
            print(f"      segments touched: {segs[:20]} ({segs.size} of {offs.size - 1})")
            ok = False
    return ok


def main():
    allok = True
    for (B, H, W), regime, fps in CASES:
        vox = synth.synthetic_voxels(B, H, W, seed=B * 100 + H, regime=regime)
        want = O.emit_soa(vox, fps=fps, seed=99, frame_base=5)
        y = torch.from_numpy(vox).cuda()
        for path, layout in (("bucket", "packed"), ("bucket", "soa"), ("sweep", "soa")):
            if path == "sweep" and fps < 12:
                continue
            tag = f"{B}x{H}x{W} {regime} fps{fps} {path}/{layout}"
            try:
                ev = ldati_device(y, fps=fps, seed=99, frame_base=5, path=path, layout=layout)
                torch.cuda.synchronize()
                ev.check()
                ok = compare(ev, want, tag)
            except Exception as e:                                  # noqa: BLE001
                print(f"  {tag}: EXCEPTION {type(e).__name__}: {e}")
                ok = False
            print(("ok   " if ok else "FAIL ") + tag + f"  events={int(want[0].sum())}")
            allok &= ok
    if "--time" in sys.argv:
        for regime in ("stress", "sparse"):
            vox = torch.from_numpy(synth.synthetic_voxels(24, 260, 346, seed=7, regime=regime)).cuda()
            for _ in range(3):
                ev = ldati_device(vox, fps=30, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ev = ldati_device(vox, fps=30, seed=1)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            nb = 80 * 24 * 260 * 346 + 13 * ev.num_events
            print(f"time {regime}: {dt * 1e3:.3f} ms per 24-pair chunk (count+sync+emit), {ev.num_events} events, "
                  f"{nb / dt / 1e9:.0f} GB/s algorithmic = {nb / dt / 8e12:.3f} of 8 TB/s")
    print("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
