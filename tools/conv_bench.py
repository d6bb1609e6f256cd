#!/usr/bin/env python3
"""Micro-benchmark of individual V2ce3d conv layers (b = 4 sequences, 346x260 pyramid) through the
C ABI; used for kernel A/B runs (tile shapes, build knobs, V2CE_DBG ablations)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2ce_toolbox_amd import hip  # noqa: E402

L = [(260, 346), (130, 173), (65, 87), (33, 44), (17, 22)]
LAYERS = {   # name: (C0, lvl0, C1, lvl_in, Cout, ksize, stride)
    "enc0.conv2": (64, 1, 0, 1, 64, 3, 1), "enc1.conv1": (64, 1, 0, 1, 128, 3, 2),
    "res0.conv1": (512, 4, 0, 4, 512, 3, 1), "enc0.conv1": (32, 0, 0, 0, 64, 3, 2), "enc3.conv1": (256, 3, 0, 3, 512, 3, 2), "dec1.conv1": (256, 3, 128, 2, 128, 3, 1),
    "dec2.conv1": (128, 2, 64, 1, 64, 3, 1), "dec0.conv1": (512, 4, 256, 3, 256, 3, 1), "dec1.conv2": (128, 2, 0, 2, 128, 3, 1),
    "enc2.conv1": (128, 2, 0, 2, 256, 3, 2), "enc2.conv2": (256, 3, 0, 3, 256, 3, 1), "enc1.conv2": (128, 2, 0, 2, 128, 3, 1),
    "dec3.conv1": (64, 1, 32, 0, 32, 3, 1), "dec3.conv2": (32, 0, 0, 0, 32, 3, 1),
    "dec2.down": (128, 2, 64, 1, 64, 1, 1), "dec1.down": (256, 3, 128, 2, 128, 1, 1), "res0.down": (512, 4, 0, 4, 512, 1, 1),
    "enc1.down": (64, 1, 0, 1, 128, 1, 2), "enc3.down": (256, 3, 0, 3, 512, 1, 2), "dec3.down": (64, 1, 32, 0, 32, 1, 1), "pred": (32, 0, 0, 0, 20, 1, 1), "enc0.down": (32, 0, 0, 0, 64, 1, 2),
}


def run(name, B=4, T=16, iters=5, tile=(0, 0, 0)):
    C0, l0, C1, lin, Cout, k, s = LAYERS[name]
    H0, W0 = L[l0]
    Hin, Win = L[lin]
    dev = "cuda"
    x0 = torch.randn(B, T, C0, H0, W0, device=dev)
    x1 = torch.randn(B, T, C1, Hin, Win, device=dev) if C1 else None
    hmap = wmap = None
    if (H0, W0) != (Hin, Win):
        hmap = (torch.arange(Hin, device=dev) * H0 // Hin).int()
        wmap = (torch.arange(Win, device=dev) * W0 // Win).int()
    pad = k // 2
    Ho, Wo = (Hin + 2 * pad - k) // s + 1, (Win + 2 * pad - k) // s + 1
    split = os.environ.get("PRECISION", "f32") == "f16x2" and Cout % 32 == 0 and (C0 + C1) % 16 == 0
    if split:
        w32 = torch.randn(Cout, C0 + C1, k, k, k, device=dev) * 0.02
        w = torch.empty(2 * w32.numel() + 4, dtype=torch.float16, device=dev)
        hip.check(hip.lib().v2ce_pack_weights_f16x2(w32.data_ptr(), Cout, C0 + C1, k ** 3, None, w.data_ptr(),
                                                    hip.stream_ptr()), "pack")
    else:
        w = torch.randn((C0 + C1) * k ** 3 * Cout, device=dev) * 0.02
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    y = torch.empty(B, T, Cout, Ho, Wo, device=dev)
    d = hip.ConvDesc(B=B, T=T, C0=C0, H0=H0, W0=W0, C1=C1, Hin=Hin, Win=Win, Cout=Cout, Hout=Ho, Wout=Wo,
                     ksize=k, stride_hw=s, act=1, tile_t=tile[0], tile_h=tile[1], tile_w=tile[2],
                     precision=1 if split else 0, layout=1 if split else 0)   # same bytes, random data: only the layout flag differs
    lib = hip.lib()

    fuse = os.environ.get("FUSE", "") if split and k == 3 else ""
    if fuse == "sc":           # the block's 1x1x1 shortcut on the same launch (v2ce_conv3d_fwd_sc)
        wd32 = torch.randn(Cout, C0 + C1, 1, 1, 1, device=dev) * 0.1
        wd = torch.empty(2 * wd32.numel() + 4, dtype=torch.float16, device=dev)
        hip.check(lib.v2ce_pack_weights_f16x2(wd32.data_ptr(), Cout, C0 + C1, 1, None, wd.data_ptr(), hip.stream_ptr()), "pack")
        ysc = torch.empty_like(y)
    if fuse == "pred":         # the 1x1x1 head on the same launch (v2ce_conv3d_fwd_pred)
        wp32 = torch.randn(20, 32, device=dev) * 0.1
        tab = torch.empty(lib.v2ce_pack_pred_weights_f16x2_bytes() // 2, dtype=torch.float16, device=dev)
        hip.check(lib.v2ce_pack_pred_weights_f16x2(wp32.data_ptr(), 20, 32, tab.data_ptr(), hip.stream_ptr()), "pack")
        pb = torch.zeros(32, device=dev)
        yp = torch.empty(B, T, 20, Ho, Wo, device=dev)
        res = torch.randn_like(y)

    res_plain = torch.randn_like(y) if os.environ.get("RES") else None      # RES=1: with a residual (the blocks' conv2)
    track = bool(os.environ.get("TRACK"))                                    # TRACK=1: range tracking like in the network
    ax0 = x0.abs().max().reshape(1) if track else None
    ax1 = x1.abs().max().reshape(1) if track and x1 is not None else None
    ay = torch.zeros(2, device=dev) if track else None

    def call():
        if fuse == "sc":
            hip.check(lib.v2ce_conv3d_fwd_sc(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1), hip.ptr(hmap), hip.ptr(wmap),
                                             w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), hip.ptr(ax0), hip.ptr(ax1), hip.ptr(ay),
                                             wd.data_ptr(), sc.data_ptr(), sh.data_ptr(), ysc.data_ptr(), hip.stream_ptr()), "conv_sc")
        elif fuse == "pred":
            hip.check(lib.v2ce_conv3d_fwd_pred(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1), hip.ptr(hmap), hip.ptr(wmap),
                                               w.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), None, hip.ptr(ax0), hip.ptr(ax1), hip.ptr(ay),
                                               tab.data_ptr(), pb.data_ptr(), 20, yp.data_ptr(), hip.stream_ptr()), "conv_pred")
        else:
            hip.check(lib.v2ce_conv3d_fwd(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1), hip.ptr(hmap), hip.ptr(wmap),
                                          w.data_ptr(), sc.data_ptr(), sh.data_ptr(), hip.ptr(res_plain), y.data_ptr(),
                                          hip.ptr(ax0), hip.ptr(ax1), hip.ptr(ay), hip.stream_ptr()), "conv")
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * T * Ho * Wo * Cout * (C0 + C1) * k ** 3
    has_res = fuse == 'pred' or (fuse == '' and res_plain is not None)
    return hip.conv_variant(d, hmap is not None, {'sc': 2, 'pred': 1}.get(fuse, 0) + (4 if has_res else 0)), ms, fl / ms / 1e9


if __name__ == "__main__":
    names = sys.argv[1:] or list(LAYERS)
    for n in names:
        v, ms, tf = run(n)
        print(f"{n:12s} {v:34s} {ms:8.3f} ms {tf:7.1f} TF")
