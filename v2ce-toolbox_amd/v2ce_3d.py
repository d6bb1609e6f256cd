"""V2ce3d (stage 1) -- host side mirroring ``/root/reference/scripts/v2ce_3d.py``.

``V2ce3d`` is an ``nn.Module`` whose parameter/buffer tree reproduces the reference state_dict
exactly (218 keys; ``scripts/v2ce_3d.py:13-24``, ``scripts/unet_2layer.py:203-318``,
``scripts/submodules.py:85-124,216-264``, ``scripts/spectral_norm.py:43-59``), so
``model.load_state_dict(torch.load('weights/v2ce_3d.pt'))``, ``.eval()``, ``.to('cuda')`` and
``model(x)`` work as in ``v2ce.py:30-43,81-82``.  The forward pass contains no torch compute ops:
every convolution (with its folded BatchNorm, activation, residual add, nearest-upsample + concat
input) is one ``v2ce_conv3d_fwd`` launch, the 12 spectral-norm layers one ``v2ce_sn_update_batch`` (power
iteration + split-half re-pack; include/v2ce_hip.h) in line in front of the head convolution.  ``precision`` selects the arithmetic of the residual-block convs: "f16x2" (default) =
f32 operands split into two fp16 halves on the fp16 MFMA with device-side range tracking
(f32-equivalent accuracy), "f32" = exact f32 MFMA.  Inference only (the
reference runs it under ``torch.no_grad()`` in eval mode, ``v2ce.py:41,66``).
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import hip

BN_EPS = 1e-5
BASE, NUM_ENC, NUM_RES = 32, 4, 2


class _Conv(nn.Module):
    """Holds nn.Conv3d-shaped parameters (``weight`` [, ``bias``])."""

    def __init__(self, cin, cout, k, bias):
        super().__init__()
        w = torch.empty(cout, cin, k, k, k)
        nn.init.kaiming_normal_(w, 10.0)                       # unet_2layer.py:259
        self.weight = nn.Parameter(w, requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        else:
            self.register_parameter("bias", None)


class _SNConvInner(nn.Module):
    """The ``module`` of a SpectralNorm wrapper: weight_u / weight_v / weight_bar
    (spectral_norm.py:43-59)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        w = torch.empty(cout, cin, k, k, k)
        nn.init.kaiming_normal_(w, 10.0)
        u = torch.randn(cout)
        v = torch.randn(cin * k ** 3)
        self.weight_u = nn.Parameter(u / (u.norm() + 1e-12), requires_grad=False)
        self.weight_v = nn.Parameter(v / (v.norm() + 1e-12), requires_grad=False)
        self.weight_bar = nn.Parameter(w, requires_grad=False)


class _SNConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.module = _SNConvInner(cin, cout, k)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _ConvLayer3D(nn.Module):
    """submodules.py:85-124 with norm=None: conv3d (+bias) then activation."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv3d = _Conv(cin, cout, k, bias=True)


class _ResidualBlock3D(nn.Module):
    """submodules.py:216-264 with norm='BN'."""

    def __init__(self, cin, cout, stride_hw, sn):
        super().__init__()
        self.stride_hw, self.sn, self.cin, self.cout = stride_hw, sn, cin, cout
        self.conv1 = _SNConv(cin, cout, 3) if sn else _Conv(cin, cout, 3, bias=False)
        self.bn1 = _BN(cout)
        self.bn2 = _BN(cout)
        self.conv2 = _SNConv(cout, cout, 3) if sn else _Conv(cout, cout, 3, bias=False)
        self.downsample = nn.Sequential(OrderedDict([("0", _Conv(cin, cout, 1, bias=True)),
                                                     ("1", _BN(cout))]))


class _UNet3D(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.head = _ConvLayer3D(cin, BASE, 3)
        self.encoders = nn.ModuleList(
            [_ResidualBlock3D(BASE * 2 ** i, BASE * 2 ** (i + 1), 2, sn=False) for i in range(NUM_ENC)])
        cmax = BASE * 2 ** NUM_ENC
        self.resblocks = nn.ModuleList(
            [_ResidualBlock3D(cmax, cmax, 1, sn=True) for _ in range(NUM_RES)])
        dec_in = [BASE * 2 ** (i + 1) for i in range(NUM_ENC)][::-1]          # 512,256,128,64
        self.decoders = nn.ModuleList(
            [_ResidualBlock3D(int(1.5 * c), c // 2, 1, sn=True) for c in dec_in])
        self.pred = _ConvLayer3D(BASE, cout, 1)


def _nearest_is_half(n_in: int, n_out: int) -> bool:
    """ATen's nearest source index equals dst >> 1 for this size pair (true for every n_in = ceil(n_out / 2) that was
    checked, but it is a property of a float computation: tested, not assumed)."""
    return bool(np.array_equal(_nearest_map(n_in, n_out), np.arange(n_out) >> 1))


def _nearest_map(n_in: int, n_out: int) -> np.ndarray:
    """ATen nearest: src = min(floorf(dst * (float)in/out), in-1) (unet_2layer.py:360)."""
    scale = np.float32(n_in) / np.float32(n_out)
    src = np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(src, n_in - 1).astype(np.int32)


class V2ce3d(nn.Module):
    """Drop-in for ``scripts/v2ce_3d.py:12``: ``V2ce3d()(x[B,L,2,H,W]) -> [B,L,20,H,W]``."""

    def __init__(self, in_channels=2, out_channels=20, precision: str = "f16x2", guard: str = "call"):
        """precision of the 3x3x3 convolutions of the residual blocks (99 % of the FLOP):
        "f16x2" (default): every f32 operand is split into two fp16 halves (22 bits, power-of-two
            pre-scales tracked on the device) and each product is three fp16 MFMAs accumulated in f32;
            measured against an f64 evaluation of the network its error equals the exact-f32 path's
            (profiles/r01_d_precision_report.json: rms 9.8e-8 both, 10x inside the 1e-5 parity bar);
        "f32": exact f32 MFMA arithmetic (v_mfma_f32_32x32x2_f32) everywhere.
        guard (f16x2 only) -- who enforces the f32 contract of the reference's ``model(x)`` (v2ce_3d.py:26-30) when
        a tensor's dynamic range exceeds what one scale per sequence covers (the range guard):
        "call" (default): every forward reads its own guard bound back (4 bytes, one synchronisation -- the
            reference's callers follow the call with ``.cpu()`` anyway) and, if it is above RANGE_GUARD_LIMIT,
            rewinds the spectral-norm state and repeats THAT call on the exact-f32 kernels: a bare ``model(x)`` is
            then either provably within 2.5e-6 of exact f32 arithmetic or exact f32 itself;
        "deferred": forward never synchronises; the owner of a whole clip checks ``range_guard_value()`` once and
            repeats the clip (glue.run_guarded: the CLI, pipeline.run_clip callers, bench.py)."""
        super().__init__()
        if precision not in ("f32", "f16x2"):
            raise ValueError(f"precision must be 'f32' or 'f16x2', got {precision!r}")
        if guard not in ("call", "deferred"):
            raise ValueError(f"guard must be 'call' or 'deferred', got {guard!r}")
        self.precision, self.guard = precision, guard
        self.guard_reruns = 0      # forwards repeated on the exact-f32 kernels by the per-call guard
        self.in_channels, self.out_channels = in_channels, out_channels
        self.UNet = _UNet3D(in_channels, out_channels)
        self._prep = None          # device-side derived constants (packed weights, folded BN)
        self._maps = {}
        self.calls = 0             # number of forward passes = spectral-norm iterations applied
        self.profile = None        # list -> every conv launch appends (variant, flops, ev0, ev1)

    # ---- state handling ---------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._prep = None
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._prep = None
        self._maps = {}
        self._sn_plist = self._sn_bufs = None
        return r

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("V2ce3d (HIP) is inference-only; the reference runs it in eval mode")
        return super().train(False)

    # ---- derived constants ------------------------------------------------------------------
    @staticmethod
    def _fold_bn(bn: _BN, conv_bias=None):
        scale = bn.weight / torch.sqrt(bn.running_var + BN_EPS)
        shift = bn.bias - bn.running_mean * scale
        if conv_bias is not None:
            shift = shift + conv_bias * scale
        return scale.float().contiguous(), shift.float().contiguous()

    def _pack(self, w, sigma=None, out=None, split=False):
        cout, cin = w.shape[0], w.shape[1]
        k3 = w.shape[2] * w.shape[3] * w.shape[4]
        if split and out is not None and getattr(out, "up_c0", 0):
            # decoder conv1: plain planes + the phase-folded region of the upsampled channels (v2ce_conv3d_fwd_up2)
            hip.check(hip.lib().v2ce_pack_weights_f16x2_up(w.data_ptr(), cout, out.up_c0, cin - out.up_c0, hip.ptr(sigma),
                                                           out.data_ptr(), hip.stream_ptr(w.device)),
                      "v2ce_pack_weights_f16x2_up")
            return out
        if split and out is not None and getattr(out, "wt", False):
            # Winograd F(2,3) along T: the transformed weights of W / sigma (v2ce_conv3d_fwd_wt); `ci0`: of input channels ci0.. only
            ci0 = getattr(out, "ci0", 0)
            hip.check(hip.lib().v2ce_pack_weights_f16x2_wt_slice(w.data_ptr(), cout, cin, ci0, cin - ci0, hip.ptr(sigma), out.data_ptr(),
                                                                 hip.stream_ptr(w.device)),
                      "v2ce_pack_weights_f16x2_wt_slice")
            return out
        if split:      # fp16 hi/lo planes for the split-half conv path
            if out is None:
                out = self._split_buffer(cout, cin, k3, w.device)
            hip.check(hip.lib().v2ce_pack_weights_f16x2(w.data_ptr(), cout, cin, k3, hip.ptr(sigma),
                                                        out.data_ptr(), hip.stream_ptr(w.device)),
                      "v2ce_pack_weights_f16x2")
            return out
        if out is None:
            out = torch.empty(cin * k3 * cout, dtype=torch.float32, device=w.device)
        hip.check(hip.lib().v2ce_pack_weights(w.data_ptr(), cout, cin, k3, hip.ptr(sigma),
                                              out.data_ptr(), hip.stream_ptr(w.device)),
                  "v2ce_pack_weights")
        return out

    @staticmethod
    def _split_buffer(cout, cin, k3, dev, up_c0=0, wt=False):
        """fp16 hi/lo planes + the {max |w|, pre-scale} tail of v2ce_pack_weights_f16x2; up_c0 > 0: followed by the
        phase-folded region of the first up_c0 input channels (v2ce_pack_weights_f16x2_up); wt: the Winograd-T planes of
        v2ce_pack_weights_f16x2_wt instead (36 tap slots)."""
        if wt:
            buf = torch.empty(hip.lib().v2ce_pack_weights_f16x2_wt_bytes(cout, cin) // 2, dtype=torch.float16, device=dev)
            buf.wt = True
            return buf
        if up_c0:
            buf = torch.empty(hip.lib().v2ce_pack_weights_f16x2_up_bytes(cout, up_c0, cin - up_c0) // 2, dtype=torch.float16,
                              device=dev)
            buf.up_c0 = up_c0
            return buf
        return torch.empty(hip.lib().v2ce_pack_weights_f16x2_bytes(cout, cin, k3) // 2, dtype=torch.float16,
                           device=dev)

    def _prepare(self):
        dev = self.UNet.head.conv3d.weight.device
        if dev.type != "cuda":
            raise hip.V2ceHipError("V2ce3d: parameters must be on a HIP device (.to('cuda')); "
                                   "there is no CPU path")
        for p in self.parameters():
            if p.dtype != torch.float32:
                raise hip.V2ceHipError("V2ce3d: parameters must be float32")
        P = {}
        ones = lambda c: torch.ones(c, dtype=torch.float32, device=dev)
        h = self.UNet.head.conv3d
        P["head"] = (self._pack(h.weight.contiguous()), ones(h.weight.shape[0]), h.bias.float().contiguous())
        P["head_split"] = None
        if self.precision == "f16x2" and tuple(h.weight.shape) == (32, 2, 3, 3, 3) and os.environ.get("V2CE_HEAD_SPLIT", "1") != "0":
            # the head in split-half arithmetic like every other layer (v2ce_conv3d_head_f16x2: a stream of 1 KB stores)
            tab = torch.empty(hip.lib().v2ce_pack_head_weights_f16x2_bytes() // 2, dtype=torch.float16, device=dev)
            hip.check(hip.lib().v2ce_pack_head_weights_f16x2(h.weight.contiguous().data_ptr(), tab.data_ptr(), hip.stream_ptr(dev)),
                      "v2ce_pack_head_weights_f16x2")
            P["head_split"] = (tab, h.bias.float().contiguous())
        pr = self.UNet.pred.conv3d
        P["pred"] = (self._pack(pr.weight.contiguous()), ones(pr.weight.shape[0]), pr.bias.float().contiguous())
        P["pred_fused"] = None
        if self.precision == "f16x2" and pr.weight.shape[1] == 32 and pr.weight.shape[0] <= 32:
            # the 1x1x1 head rides on the last decoder conv's accumulators (v2ce_conv3d_fwd_pred)
            tab = torch.empty(hip.lib().v2ce_pack_pred_weights_f16x2_bytes() // 2, dtype=torch.float16, device=dev)
            hip.check(hip.lib().v2ce_pack_pred_weights_f16x2(pr.weight.contiguous().data_ptr(), pr.weight.shape[0], 32,
                                                             tab.data_ptr(), hip.stream_ptr(dev)),
                      "v2ce_pack_pred_weights_f16x2")
            bias = torch.zeros(32, dtype=torch.float32, device=dev)
            bias[:pr.weight.shape[0]] = pr.bias.float()
            P["pred_fused"] = (tab, bias, int(pr.weight.shape[0]))
        sn_ws = 0
        for name, blocks in (("enc", self.UNet.encoders), ("res", self.UNet.resblocks),
                             ("dec", self.UNet.decoders)):
            for i, blk in enumerate(blocks):
                d = {}
                splits = {"conv1": self._split(blk.cin, blk.cout), "conv2": self._split(blk.cout, blk.cout)}
                d["bn1"] = self._fold_bn(blk.bn1)
                d["bn2"] = self._fold_bn(blk.bn2)
                d["down_w"] = self._pack(blk.downsample[0].weight.contiguous(),
                                         split=self._fuse_shortcut(blk) or self._split(blk.cin, blk.cout, 1, blk.stride_hw))
                d["down_bn"] = self._fold_bn(blk.downsample[1], blk.downsample[0].bias)
                d["fold"] = None
                if self._fold_shortcut(blk) or self._dec_last_split(name, i, blk):
                    # Wd' = Wd sd / s2 (one f32 rounding per weight on top of the 22-bit split: ~2^-22 relative on the shortcut's
                    # contribution); a BatchNorm scale of (almost) zero in bn2 cannot be divided out: that block keeps its own launch
                    s2, sh2 = d["bn2"]
                    sd, shd = d["down_bn"]
                    # (V2CE_WT_UNFOLD=1: keep the shortcut's own launch so that conv2 can run on the Winograd-T kernel)
                    unfold = self._winograd() and blk.cout % 64 == 0 and os.environ.get("V2CE_WT_UNFOLD", "0") == "1"
                    if not unfold and float(s2.abs().min()) > 1e-20 and bool(torch.isfinite(sd / s2).all()):
                        wf = (blk.downsample[0].weight * (sd / s2).view(-1, 1, 1, 1, 1)).contiguous()
                        d["fold"] = (self._pack(wf, split=True), s2, (sh2 + shd).contiguous())
                        d["fold_w32"] = wf
                # Winograd F(2,3) along T (v2ce_conv3d_fwd_wt) where a conv is a plain one-source stride-1 3x3x3 launch with
                # >= 64 output channels: conv2 of a block whose shortcut does not ride in its K loop, conv1 of the middle blocks
                wt = {"conv1": self._winograd() and splits["conv1"] and name == "res" and blk.stride_hw == 1 and blk.cout % 64 == 0
                      and not self._fuse_shortcut(blk),
                      "conv2": self._winograd() and splits["conv2"] and blk.cout % 64 == 0 and
                      (d["fold"] is None or (blk.cin % 64 == 0 and os.environ.get("V2CE_WT_TAIL", "1") != "0"))}
                d["wt"] = wt
                d["fold_lo"] = d["fold_skip"] = None
                if name == "dec" and d["fold"] is not None and self._upfold() and os.environ.get("V2CE_TAIL_LOWRES", "1") != "0" and \
                        ((wt["conv2"] and (blk.cin // 3) % 64 == 0) or self._dec_last_split(name, i, blk)):
                    # the folded shortcut split by source: the upsampled channels' share s2 Wd'[:, :C0] x0 is computed at the SOURCE's
                    # resolution (a 1x1x1 launch on a quarter of the positions) and added as an upsampled residual
                    # (v2ce_conv3d_fwd_wt_tail, res_h); only the skip channels ride as the tail -- a third of its gathers and MFMAs
                    c0 = blk.cin * 2 // 3
                    wf = (blk.downsample[0].weight * (d["down_bn"][0] / d["bn2"][0]).view(-1, 1, 1, 1, 1))
                    d["fold_lo"] = self._pack(wf[:, :c0].contiguous(), split=True)
                    d["fold_skip_w32"] = wf[:, c0:].contiguous()
                    d["fold_skip"] = self._pack(d["fold_skip_w32"], split=True)
                d["zero_shift"] = torch.zeros(blk.cout, dtype=torch.float32, device=dev)
                if blk.sn:
                    for cn in ("conv1", "conv2"):
                        m = getattr(blk, cn).module
                        rows, cols = m.weight_bar.shape[0], m.weight_bar[0].numel()
                        # a decoder's conv1 reads upsample(x) ++ skip: its first 2/3 input channels are packed phase-folded too
                        up_c0 = blk.cin * 2 // 3 if (name == "dec" and cn == "conv1" and self._upfold()) else 0
                        d[cn + "_w"] = (self._split_buffer(rows, m.weight_bar.shape[1], 27, dev, up_c0=up_c0, wt=wt[cn])
                                        if splits[cn] else torch.empty(rows * cols, dtype=torch.float32, device=dev))
                        if up_c0 and self._up_split(blk):
                            # the skip channels of a wide decoder's conv1 on the Winograd-T kernel (v2ce_conv3d_fwd_up2_part + v2ce_conv3d_fwd_wt)
                            sk = self._split_buffer(rows, m.weight_bar.shape[1] - up_c0, 27, dev, wt=True)
                            sk.ci0 = up_c0
                            d["conv1_skip_w"] = sk
                            d["zero_shift"] = torch.zeros(rows, dtype=torch.float32, device=dev)
                        sn_ws = max(sn_ws, hip.lib().v2ce_sn_workspace_bytes(rows, cols))
                else:
                    for cn in ("conv1", "conv2"):
                        w = getattr(blk, cn).weight.contiguous()
                        out = self._split_buffer(w.shape[0], w.shape[1], 27, dev, wt=True) if wt[cn] else None
                        d[cn + "_w"] = self._pack(w, out=out, split=splits[cn])
                P[f"{name}{i}"] = d
        P["sn_ws"] = torch.empty(max(sn_ws, 16), dtype=torch.uint8, device=dev)
        P["sigma"] = torch.empty(1, dtype=torch.float32, device=dev)
        P["sn_batch"] = None
        if self.precision == "f16x2":
            # all 12 spectral-norm layers in one v2ce_sn_update_batch call (six launches instead of 84)
            inners = [(getattr(blk, cn).module, P[f"{name}{i}"][cn + "_w"], P[f"{name}{i}"].get("conv1_skip_w") if cn == "conv1" else None)
                      for name, blocks in (("res", self.UNet.resblocks), ("dec", self.UNet.decoders))
                      for i, blk in enumerate(blocks) if blk.sn for cn in ("conv1", "conv2")]
            arr = (hip.SnLayer * len(inners))()
            for e, (m, out, skip) in zip(arr, inners):
                e.w_bar, e.u, e.v, e.packed = m.weight_bar.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr(), out.data_ptr()
                e.rows, e.cols, e.k3 = m.weight_bar.shape[0], m.weight_bar[0].numel(), 27
                e.up_c0 = getattr(out, "up_c0", 0)
                e.wt = 1 if getattr(out, "wt", False) else 0
                e.packed_skip = None if skip is None else skip.data_ptr()
            nb = hip.lib().v2ce_sn_batch_workspace_bytes(arr, len(inners))
            if nb:
                P["sn_batch"] = (arr, len(inners), torch.empty(nb, dtype=torch.uint8, device=dev))
            P["sn_once"] = None
            if nb and os.environ.get("V2CE_SN_REPACK", "0") != "1":
                # Round 6: W_bar is constant, only the scalar sigma changes from call to call (spectral_norm.py:31: w = W_bar / sigma).
                # The planes of W_bar -- plain, phase-folded, Winograd-T: all linear in W_bar -- are packed ONCE, here, and 1 / sigma
                # rides in the launch's epilogue scale (bn scale / sigma, written by the batched power iteration); a block's folded
                # shortcut Wd' sits in the same accumulators and is therefore re-packed as Wd' sigma (five 1x1x1 tensors, 0.8 M
                # weights instead of 38 M).  One f32 rounding moves from every weight to every output: ~1e-7 relative.
                # V2CE_SN_REPACK=1: the re-pack of every forward (rounds 1-5).
                one = torch.ones(1, dtype=torch.float32, device=dev)
                inv_sigma = torch.ones(len(inners), dtype=torch.float32, device=dev)
                keep, tails, k = [one, inv_sigma], [], 0
                for name, blocks in (("res", self.UNet.resblocks), ("dec", self.UNet.decoders)):
                    for i, blk in enumerate(blocks):
                        if not blk.sn:
                            continue
                        d = P[f"{name}{i}"]
                        for cn, bn in (("conv1", "bn1"), ("conv2", "bn2")):
                            m, out, skip = inners[k]
                            self._pack(m.weight_bar, one, out, split=True)
                            if skip is not None:
                                self._pack(m.weight_bar, one, skip, split=True)
                            eff = torch.empty_like(d[bn][0])
                            d[bn + "_eff"] = (eff, d[bn][1])
                            e = arr[k]
                            e.flags = hip.SN_NO_PACK
                            e.bn_scale, e.scale_out = d[bn][0].data_ptr(), eff.data_ptr()
                            e.inv_sigma_out = inv_sigma[k:].data_ptr()
                            if cn == "conv2" and d.get("fold") is not None:
                                for w32, buf in ((d["fold_w32"], d["fold"][0]), (d.get("fold_skip_w32"), d.get("fold_skip"))):
                                    if w32 is not None:
                                        tails.append((w32, buf, k, float(w32.abs().max())))
                            k += 1
                tarr = (hip.SnLayer * max(len(tails), 1))()
                for e, (w32, buf, kk, wmax) in zip(tarr, tails):
                    e.w_bar, e.packed = w32.data_ptr(), buf.data_ptr()
                    e.rows, e.cols, e.k3 = w32.shape[0], w32.shape[1], 1
                    e.flags, e.sigma_src, e.wmax = hip.SN_NO_ITERATE, inv_sigma[kk:].data_ptr(), wmax
                tws = None
                if tails:
                    tws = torch.empty(max(hip.lib().v2ce_sn_batch_workspace_bytes(tarr, len(tails)), 16), dtype=torch.uint8, device=dev)
                P["sn_once"] = {"one": one, "inv_sigma": inv_sigma, "tails": tails, "tail_batch": (tarr, len(tails), tws), "keep": keep}
        # one max-|y| slot per conv launch of a forward pass (split-half path: the consumer derives its
        # power-of-two activation pre-scale from the producer's slot, all on the device)
        # (+ the launch's range-guard value in the second float, include/v2ce_hip.h)
        # -- per BATCH ELEMENT (desc.absmax_batch_stride = 2): a sequence's pre-scales, and with them its result, do
        # not depend on what else is in the batch; sized per batch size in _forward
        P["absmax"] = torch.zeros((64, 1, 2), dtype=torch.float32, device=dev)
        P["guard"] = torch.zeros(1, dtype=torch.float32, device=dev)     # max guard value since the last read
        self._prep = P

    def _split(self, cin, cout, ksize=3, stride=1) -> bool:
        """Split-half arithmetic for a conv of a residual block?  All of them have Cin % 16 == 0 and
        Cout % 32 == 0.  Every 3x3x3 conv; of the 1x1x1 shortcuts the strided ones and those with
        >= 64 output channels (measured: >= 128 channels 0.21 vs 0.35 ms, strided 0.12 vs 0.35, the 192 -> 64
        one 0.40 vs 0.53 since the epilogue rework); only a 32-channel stride-1 shortcut (fused into conv1
        in this network anyway) is faster on the exact-f32 kernel."""
        if self.precision != "f16x2":
            return False
        return ksize == 3 or stride == 2 or cout >= 64

    def _winograd(self) -> bool:
        """Winograd F(2,3) along T for the plain stride-1 3x3x3 convs (v2ce_conv3d_fwd_wt: 2/3 of the multiplies;
        V2CE_WINOGRAD=0: the direct kernel, for A/B runs)."""
        return self.precision == "f16x2" and os.environ.get("V2CE_WINOGRAD", "1") != "0"

    def _up_split(self, blk) -> bool:
        """conv1 of a decoder with >= 64 output channels as TWO launches: the upsampled channels phase-folded
        (v2ce_conv3d_fwd_up2_part, no activation) and the skip channels on the Winograd-T kernel with the first launch's output as
        its residual -- the skip channels are 53 % of the layer's multiplies, two thirds of them remain.  V2CE_UP_SPLIT=0: one launch."""
        # (the shallower the block the larger the second launch's epilogue and the partial sum's round trip against the multiplies
        # saved: measured per block in tools/upsplit_ab.sh -- dec0 1.74 -> 1.52 ms, dec1 1.51 -> 1.49, dec2 1.47 -> 1.50)
        return self._upfold() and self._winograd() and blk.cout % 64 == 0 and os.environ.get("V2CE_UP_SPLIT", "1") != "0" and \
            blk.cout >= int(os.environ.get("V2CE_UP_SPLIT_MIN_COUT", "128"))

    def _upfold(self) -> bool:
        """Phase-folded upsampled channels in the decoders' conv1 (v2ce_conv3d_fwd_up2; V2CE_UPFOLD=0: the mapped gather
        with all 27 taps, for A/B runs)."""
        return self.precision == "f16x2" and os.environ.get("V2CE_UPFOLD", "1") != "0"

    def _fuse_shortcut(self, blk) -> bool:
        """Ride the block's 1x1x1 shortcut on conv1's launch (v2ce_conv3d_fwd_sc)?  Possible where conv1's
        waves own one 32-channel fragment row (strided blocks, and the 32-channel last decoder): the
        second accumulator set fits, the shortcut's input is never read a second time and its launch
        disappears (enc0-3, dec3: ~0.9 ms per 64 frame-pairs)."""
        if self.precision != "f16x2":
            return False
        if blk.stride_hw == 2 and os.environ.get("V2CE_FOLD_STRIDED", "0") == "1":
            return False                                   # A/B: the strided blocks' shortcuts as conv2 tails (stride-2 gather)
        return blk.stride_hw == 2 or blk.cout <= 32

    def _dec_last_split(self, name, i, blk) -> bool:
        """Round 6: the LAST decoder block (32 channels; the `pred` head rides on its conv2) with its 1x1x1 shortcut split by source
        like dec0-2's -- the upsampled channels' share at the source's resolution as conv2's low-resolution residual, the skip channels as
        conv2's folded tail (v2ce_conv3d_fwd_tail_pred) -- instead of riding on conv1 as a second accumulator set that writes a second
        full-resolution tensor and leaves conv1 no registers to double-buffer its B fragments.
        OPT-IN (V2CE_DEC3_SPLIT=1): measured on 64 frame-pairs (profiles/r06_c_dec3_split_ab.txt) conv1 1.96 -> 1.66 ms as predicted, but
        conv2 + pred 1.02 -> 1.24 ms -- its tail reads the 737 MB skip tensor that conv1 had read anyway, so the launch moves as many bytes
        as with the full-resolution shortcut tensor it no longer reads -- and the low-resolution 1x1x1 launch adds 0.15 ms (409 MB in,
        205 MB out): 16.32 -> 16.38 ms per step.  What paid for dec0-2 (whose shortcut was a launch of its own) does not pay where the
        shortcut rode on a launch that had both sources in its LDS already."""
        return (self.precision == "f16x2" and name == "dec" and i == len(self.UNet.decoders) - 1 and blk.cout == 32 and blk.stride_hw == 1
                and blk.cin == 96 and self._upfold() and os.environ.get("V2CE_DEC3_SPLIT", "0") == "1")

    def _fold_shortcut(self, blk) -> bool:
        """Fold the block's 1x1x1 shortcut into conv2's K loop (v2ce_conv3d_fwd_tail)?  Where it does not already ride on
        conv1 (``_fuse_shortcut``) and conv2 has >= 64 output channels: res0-1, dec0-2.  V2CE_FOLD_SHORTCUT=0 disables."""
        return self.precision == "f16x2" and not self._fuse_shortcut(blk) and blk.cout >= 64 and \
            os.environ.get("V2CE_FOLD_SHORTCUT", "1") != "0"

    def _map(self, n_in, n_out, dev):
        key = (n_in, n_out, str(dev))
        if key not in self._maps:
            self._maps[key] = torch.from_numpy(_nearest_map(n_in, n_out)).to(dev)
        return self._maps[key]

    # ---- kernels ------------------------------------------------------------------------------
    @staticmethod
    def to_c16(x: torch.Tensor) -> torch.Tensor:
        """Planar [B,T,C,H,W] -> channels-last-16 [B,T,C/16,H,W,16] (copy; tests and callers of the raw kernels)."""
        B, T, C, H, W = x.shape
        y = x.reshape(B, T, C // 16, 16, H, W).permute(0, 1, 2, 4, 5, 3).contiguous()
        y.lw, y.c16 = getattr(x, "lw", W), True
        if hasattr(x, "absmax"):
            y.absmax = x.absmax
        return y

    @staticmethod
    def to_planar(x: torch.Tensor) -> torch.Tensor:
        """Channels-last-16 -> planar [B,T,C,H,lw] (copy, padding columns dropped)."""
        if not getattr(x, "c16", False):
            return x[..., :getattr(x, "lw", x.shape[4])]
        B, T, G, H, Wp, _ = x.shape
        y = x.permute(0, 1, 2, 5, 3, 4).reshape(B, T, G * 16, H, Wp)[..., :getattr(x, "lw", Wp)].contiguous()
        if hasattr(x, "absmax"):
            y.absmax = x.absmax
        return y

    @staticmethod
    def _pitch(w: int) -> int:
        """Row pitch of an intermediate activation of width w: rows of at least 64 floats are padded to a multiple
        of 32 floats, so that the 32-position pieces the conv kernels store and gather are whole cache lines
        (346 -> 352, 173 -> 192, 87 -> 96; tools/micro/store_rate.hip).  V2CE_ACT_PITCH=0 disables it."""
        if w < 64 or os.environ.get("V2CE_ACT_PITCH", "1") == "0":
            return w
        return (w + 31) // 32 * 32

    def _conv(self, x0, x1, w_packed, scale, shift, cout, ksize, stride, act, residual=None,
              up_to=None, split=False, track=False, pred=None, sc=None, dense_out=False, tail=None, residual_up=False,
              algo_hw=None):
        """x0 [B,T,C0,H0,W0] (optionally nearest-upsampled to ``up_to``), x1 [B,T,C1,Hin,Win].
        ``algo_hw``: the output resolution at which the REFERENCE computes this launch's operator when the launch itself runs at the
        source's (the upsampled channels' share of a decoder shortcut, submodules.py:262 on unet_2layer.py:360's upsampled tensor):
        only the profile's algorithmic flop column uses it."""
        # activations between the layers are [B,T,C,H,pitch] (planar) or [B,T,C/16,H,pitch,16] (`.c16`: what the
        # split-half kernels take and produce, include/v2ce_hip.h V2CE_LAYOUT_C16), logical width in `.lw` (see _pitch)
        head_bridge = (not split and self.precision == "f16x2" and ksize == 3 and stride == 1 and x1 is None
                       and x0.dim() == 5 and x0.shape[2] == 2 and cout == 32 and pred is None and not dense_out)
        c16 = split or head_bridge
        if not c16 and getattr(x0, "c16", False):        # exact-f32 kernels read planar tensors (test / intermediates paths)
            x0 = self.to_planar(x0)
        if split:
            assert getattr(x0, "c16", False) and (x1 is None or getattr(x1, "c16", False)) and \
                (residual is None or getattr(residual, "c16", False)), "split-half launches take channels-last-16 activations"
        B, T, _, H0, W0p = x0.shape[:5]
        C0 = x0.shape[2] * (16 if getattr(x0, "c16", False) else 1)
        W0 = getattr(x0, "lw", W0p)
        Hin, Win = up_to if up_to is not None else (H0, W0)
        hmap = wmap = None
        if (Hin, Win) != (H0, W0):
            hmap, wmap = self._map(H0, Hin, x0.device), self._map(W0, Win, x0.device)
        # exact 2x nearest upsample (ATen's map is then dst >> 1: _nearest_is_half) into a conv whose weights carry the
        # phase-folded region: the decoder kernel (12 instead of 27 taps on the upsampled channels)
        up2 = (split and hmap is not None and x1 is not None and getattr(w_packed, "up_c0", 0) == x0.shape[2] * 16 and ksize == 3
               and stride == 1 and pred is None and tail is None and residual is None
               and H0 == (Hin + 1) // 2 and W0 == (Win + 1) // 2 and _nearest_is_half(H0, Hin) and _nearest_is_half(W0, Win))
        # Winograd F(2,3) along T (v2ce_conv3d_fwd_wt): a one-source stride-1 3x3x3 conv whose weights were packed transformed
        wt = split and getattr(w_packed, "wt", False)
        if wt:
            assert ksize == 3 and stride == 1 and x1 is None and hmap is None and pred is None and sc is None, \
                "Winograd-T weights drive only the one-source 3x3x3 stride-1 launch (plain, with a residual, or with a folded tail)"
        C1 = 0 if x1 is None else x1.shape[2] * (16 if getattr(x1, "c16", False) else 1)
        Winp = Win if x1 is None else x1.shape[4]
        assert x1 is None or getattr(x1, "lw", Winp) == Win
        pad = ksize // 2
        Hout = (Hin + 2 * pad - ksize) // stride + 1
        Wout = (Win + 2 * pad - ksize) // stride + 1
        Woutp = self._pitch(Wout) if dense_out is False else Wout
        assert residual is None or residual_up or residual.shape[4] == Woutp
        if residual_up:         # a low-resolution residual read at (h >> 1, w >> 1): the Winograd-T tail launch and the tail + head launch
            assert (wt or pred is not None) and tail is not None and residual.shape[3] == (Hout + 1) // 2 and getattr(residual, "c16", False)
        y = torch.empty((B, T, cout // 16, Hout, Woutp, 16) if c16 else (B, T, cout, Hout, Woutp),
                        dtype=torch.float32, device=x0.device)
        y.lw, y.c16 = Wout, c16
        d = hip.ConvDesc(B=B, T=T, C0=C0, H0=H0, W0=W0, C1=C1, Hin=Hin, Win=Win, Cout=cout,
                         Hout=Hout, Wout=Wout, ksize=ksize, stride_hw=stride, act=act,
                         tile_t=0, tile_h=0, tile_w=0,
                         precision=hip.PRECISION_F16X2 if split else hip.PRECISION_F32,
                         W0_pitch=W0p, Win_pitch=Winp, Wout_pitch=Woutp,
                         layout=hip.LAYOUT_C16 if c16 else hip.LAYOUT_PLANAR,
                         # [slot][B][2] table: one range slot per batch element (callers of the raw kernels that
                         # keep a [slot][2] table get one slot per tensor)
                         absmax_batch_stride=2 if (getattr(self, "_prep", None) or {}).get("absmax", torch.empty(0)).dim() == 3 else 0)
        a0 = a1 = ay = None
        if track or split:         # range tracking for the split-half consumers (device side only)
            ay = y.absmax = self._prep["absmax"][self._slot]              # [B] x [max |y|, range-guard value]
            self._slot += 1
            if split:          # untracked inputs (None) select the kernel's fixed pre-scale
                a0 = getattr(x0, "absmax", None)
                a1 = None if x1 is None or a0 is None else x1.absmax
        prof = self._prof_list()
        if prof is not None:       # HIP events on the launch stream (torch's current stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if wt and tail is None:
            hip.check(hip.lib().v2ce_conv3d_fwd_wt(ctypes.byref(d), x0.data_ptr(), w_packed.data_ptr(), scale.data_ptr(),
                                                   shift.data_ptr(), hip.ptr(residual), y.data_ptr(), hip.ptr(a0), hip.ptr(ay),
                                                   hip.stream_ptr(x0.device)),
                      "v2ce_conv3d_fwd_wt")
        elif up2:
            y_sc = None
            if sc is not None:
                y_sc = torch.empty_like(y)
                y_sc.lw, y_sc.c16 = Wout, c16
            sc_w, sc_scale, sc_shift = sc if sc is not None else (None, None, None)
            hip.check(hip.lib().v2ce_conv3d_fwd_up2(ctypes.byref(d), x0.data_ptr(), x1.data_ptr(), w_packed.data_ptr(),
                                                    scale.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                    hip.ptr(a0), hip.ptr(a1), hip.ptr(ay), hip.ptr(sc_w), hip.ptr(sc_scale),
                                                    hip.ptr(sc_shift), hip.ptr(y_sc), hip.stream_ptr(x0.device)),
                      "v2ce_conv3d_fwd_up2")
        elif pred is not None and tail is not None:      # the fused head behind a conv with a folded tail and a low-resolution residual
            tab, pbias, pcout = pred
            tx0, tx1, t_up_to, t_stride, tw = tail
            assert tx1 is None and t_up_to is None and t_stride == 1 and getattr(tx0, "c16", False) and x1 is None and hmap is None
            assert residual is None or (residual_up and residual.shape[3] == (Hout + 1) // 2 and getattr(residual, "c16", False))
            tC0, tH0, tW0p = tx0.shape[2] * 16, tx0.shape[3], tx0.shape[4]
            tW0 = getattr(tx0, "lw", tW0p)
            td = hip.ConvDesc(B=B, T=T, C0=tC0, H0=tH0, W0=tW0, C1=0, Hin=tH0, Win=tW0, Cout=cout, Hout=Hout, Wout=Wout, ksize=1,
                              stride_hw=1, act=hip.ACT_NONE, tile_t=0, tile_h=0, tile_w=0, precision=hip.PRECISION_F16X2,
                              W0_pitch=tW0p, Win_pitch=tW0p, Wout_pitch=Woutp, layout=hip.LAYOUT_C16, absmax_batch_stride=d.absmax_batch_stride)
            y = torch.empty((B, T, pcout, Hout, Wout), dtype=torch.float32, device=x0.device)
            hip.check(hip.lib().v2ce_conv3d_fwd_tail_pred(ctypes.byref(d), x0.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                          None, hip.ptr(a0), hip.ptr(ay), tab.data_ptr(), pbias.data_ptr(), pcout, y.data_ptr(),
                                                          ctypes.byref(td), tx0.data_ptr(), None, None, None, tw.data_ptr(),
                                                          hip.ptr(getattr(tx0, "absmax", None) if a0 is not None else None), None,
                                                          hip.ptr(residual), residual.shape[3] if residual is not None else 0,
                                                          residual.shape[4] if residual is not None else 0, hip.stream_ptr(x0.device)),
                      "v2ce_conv3d_fwd_tail_pred")
        elif pred is not None:           # fused 1x1x1 head: only its output is materialised
            tab, pbias, pcout = pred
            y = torch.empty((B, T, pcout, Hout, Wout), dtype=torch.float32, device=x0.device)
            hip.check(hip.lib().v2ce_conv3d_fwd_pred(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1),
                                                     hip.ptr(hmap), hip.ptr(wmap), w_packed.data_ptr(),
                                                     scale.data_ptr(), shift.data_ptr(), hip.ptr(residual),
                                                     None, hip.ptr(a0), hip.ptr(a1), hip.ptr(ay),
                                                     tab.data_ptr(), pbias.data_ptr(), pcout, y.data_ptr(),
                                                     hip.stream_ptr(x0.device)),
                      "v2ce_conv3d_fwd_pred")
        elif tail is not None:         # folded 1x1x1 tail: the block's shortcut inside this launch's K loop
            tx0, tx1, t_up_to, t_stride, tw = tail
            assert getattr(tx0, "c16", False) and (tx1 is None or getattr(tx1, "c16", False)) and (residual is None or wt)
            tC0, tH0, tW0p = tx0.shape[2] * 16, tx0.shape[3], tx0.shape[4]
            tW0 = getattr(tx0, "lw", tW0p)
            tHin, tWin = t_up_to if t_up_to is not None else (tH0, tW0)
            thmap = twmap = None
            if (tHin, tWin) != (tH0, tW0):
                thmap, twmap = self._map(tH0, tHin, tx0.device), self._map(tW0, tWin, tx0.device)
            td = hip.ConvDesc(B=B, T=T, C0=tC0, H0=tH0, W0=tW0, C1=0 if tx1 is None else tx1.shape[2] * 16, Hin=tHin, Win=tWin,
                              Cout=cout, Hout=Hout, Wout=Wout, ksize=1, stride_hw=t_stride, act=hip.ACT_NONE,
                              tile_t=0, tile_h=0, tile_w=0, precision=hip.PRECISION_F16X2, W0_pitch=tW0p,
                              Win_pitch=tWin if tx1 is None else tx1.shape[4], Wout_pitch=Woutp, layout=hip.LAYOUT_C16,
                              absmax_batch_stride=d.absmax_batch_stride)
            ta0 = getattr(tx0, "absmax", None) if a0 is not None else None
            ta1 = None if tx1 is None or ta0 is None else tx1.absmax
            if wt:
                hip.check(hip.lib().v2ce_conv3d_fwd_wt_tail(ctypes.byref(d), x0.data_ptr(), w_packed.data_ptr(), scale.data_ptr(),
                                                            shift.data_ptr(), y.data_ptr(), hip.ptr(a0), hip.ptr(ay), ctypes.byref(td),
                                                            tx0.data_ptr(), hip.ptr(tx1), hip.ptr(thmap), hip.ptr(twmap), tw.data_ptr(),
                                                            hip.ptr(ta0), hip.ptr(ta1), hip.ptr(residual),
                                                            residual.shape[3] if residual_up else 0, residual.shape[4] if residual_up else 0,
                                                            hip.stream_ptr(x0.device)),
                          "v2ce_conv3d_fwd_wt_tail")
            else:
                hip.check(hip.lib().v2ce_conv3d_fwd_tail(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1), hip.ptr(hmap), hip.ptr(wmap),
                                                         w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                         hip.ptr(a0), hip.ptr(a1), hip.ptr(ay), ctypes.byref(td), tx0.data_ptr(),
                                                         hip.ptr(tx1), hip.ptr(thmap), hip.ptr(twmap), tw.data_ptr(),
                                                         hip.ptr(ta0), hip.ptr(ta1), hip.stream_ptr(x0.device)),
                          "v2ce_conv3d_fwd_tail")
        elif sc is not None:           # fused 1x1x1 shortcut: second output tensor
            sc_w, sc_scale, sc_shift = sc
            y_sc = torch.empty_like(y)
            y_sc.lw, y_sc.c16 = Wout, c16
            hip.check(hip.lib().v2ce_conv3d_fwd_sc(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1),
                                                   hip.ptr(hmap), hip.ptr(wmap), w_packed.data_ptr(),
                                                   scale.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                   hip.ptr(a0), hip.ptr(a1), hip.ptr(ay),
                                                   sc_w.data_ptr(), sc_scale.data_ptr(), sc_shift.data_ptr(),
                                                   y_sc.data_ptr(), hip.stream_ptr(x0.device)),
                      "v2ce_conv3d_fwd_sc")
        else:
            hip.check(hip.lib().v2ce_conv3d_fwd(ctypes.byref(d), x0.data_ptr(), hip.ptr(x1),
                                                hip.ptr(hmap), hip.ptr(wmap), w_packed.data_ptr(),
                                                scale.data_ptr(), shift.data_ptr(), hip.ptr(residual),
                                                y.data_ptr(), hip.ptr(a0), hip.ptr(a1), hip.ptr(ay),
                                                hip.stream_ptr(x0.device)),
                      "v2ce_conv3d_fwd")
        if prof is not None:
            e1.record()
            flops = 2.0 * B * T * Hout * Wout * cout * (C0 + C1) * ksize ** 3
            launch_flops = flops
            if algo_hw is not None:        # a 1x1x1 launch at the source's resolution standing for the reference's at the upsampled one
                flops = 2.0 * B * T * algo_hw[0] * algo_hw[1] * cout * (C0 + C1) * ksize ** 3
            if pred is not None:
                flops += 2.0 * B * T * Hout * Wout * pred[2] * cout
            if sc is not None:
                flops += 2.0 * B * T * Hout * Wout * cout * (C0 + C1)
            if tail is not None:
                flops += 2.0 * B * T * Hout * Wout * cout * 16 * (tail[0].shape[2] + (0 if tail[1] is None else tail[1].shape[2]))
            name = hip.conv_wt_variant(d, (3 if residual is not None else 2) if tail is not None else residual is not None) if wt else hip.conv_up2_variant(d, sc is not None) if up2 else \
                hip.conv_variant(d, hmap is not None, ((8 if tail is not None else 1) if pred is not None else (2 if sc is not None else (3 if tail is not None else 0))) +
                                 (4 if residual is not None else 0))
            # (flops = the ALGORITHMIC count of the reference's convolution; a phase-folded launch executes fewer: `executed`)
            executed = flops - (2.0 * B * T * Hout * Wout * cout * C0 * 15 if up2 else 0.0)
            if algo_hw is not None:
                executed = launch_flops
            if wt:          # four transformed 3x3 convolutions per pair of time steps instead of six tap rows
                executed = 2.0 * B * ((T + 1) // 2) * Hout * Wout * cout * C0 * 36
            prof.append((name, flops, e0, e1, executed))
        if sc is not None:
            return y, y_sc
        return y

    @staticmethod
    def _half_up(x0, up_to) -> bool:
        """x0 is the source of an exact 2x nearest upsample to `up_to` (ATen's map is then dst >> 1)."""
        H0, W0 = x0.shape[3], getattr(x0, "lw", x0.shape[4])
        Hin, Win = up_to
        return H0 == (Hin + 1) // 2 and W0 == (Win + 1) // 2 and _nearest_is_half(H0, Hin) and _nearest_is_half(W0, Win)

    def _up2_ok(self, x0, x1, w_packed, up_to) -> bool:
        """The phase-folded decoder launch applies: exact 2x nearest upsample (ATen's map is then dst >> 1) into folded weights."""
        H0, W0 = x0.shape[3], getattr(x0, "lw", x0.shape[4])
        Hin, Win = up_to
        return (getattr(x0, "c16", False) and getattr(x1, "c16", False) and getattr(w_packed, "up_c0", 0) == x0.shape[2] * 16
                and H0 == (Hin + 1) // 2 and W0 == (Win + 1) // 2 and _nearest_is_half(H0, Hin) and _nearest_is_half(W0, Win))

    def _conv_up_part(self, x0, x1, w_up, scale, shift, cout, up_to):
        """scale * conv(upsample(x0); W[:, :C0]) + shift on the phase-folded kernel, no activation (v2ce_conv3d_fwd_up2_part): the
        residual of the Winograd-T launch that adds the skip channels' share and the activation."""
        B, T, G0, H0, W0p = x0.shape[:5]
        C0, C1 = G0 * 16, x1.shape[2] * 16
        W0 = getattr(x0, "lw", W0p)
        Hin, Win = up_to
        Woutp = self._pitch(Win)
        y = torch.empty((B, T, cout // 16, Hin, Woutp, 16), dtype=torch.float32, device=x0.device)
        y.lw, y.c16 = Win, True
        d = hip.ConvDesc(B=B, T=T, C0=C0, H0=H0, W0=W0, C1=C1, Hin=Hin, Win=Win, Cout=cout, Hout=Hin, Wout=Win, ksize=3, stride_hw=1,
                         act=hip.ACT_NONE, tile_t=0, tile_h=0, tile_w=0, precision=hip.PRECISION_F16X2, W0_pitch=W0p,
                         Win_pitch=x1.shape[4], Wout_pitch=Woutp, layout=hip.LAYOUT_C16,
                         absmax_batch_stride=2 if self._prep["absmax"].dim() == 3 else 0)
        ay = self._prep["absmax"][self._slot]          # (only its range-guard value matters: the partial sum feeds no split-half launch)
        self._slot += 1
        a0 = getattr(x0, "absmax", None)
        prof = self._prof_list()
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(hip.lib().v2ce_conv3d_fwd_up2_part(ctypes.byref(d), x0.data_ptr(), w_up.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                     y.data_ptr(), hip.ptr(a0), hip.ptr(a0), hip.ptr(ay), hip.stream_ptr(x0.device)),
                  "v2ce_conv3d_fwd_up2_part")
        if prof is not None:
            e1.record()
            per_tap = 2.0 * B * T * Hin * Win * cout * C0
            prof.append((hip.conv_up2_variant(d, False), 27 * per_tap, e0, e1, 12 * per_tap))
        return y

    def _prof_list(self):
        """The list this launch's HIP events go to, or None.  ``profile_filter`` (a set of launch indices of a forward call, or None):
        events only around those launches -- ~90 event records per forward cost 0.13-0.24 ms of a 16 ms step (bench.py profiles
        every launch in its warm-up steps and only the dominant kernel family in the timed ones)."""
        prof = getattr(self, "profile", None)
        idx = getattr(self, "_launch_idx", 0)
        self._launch_idx = idx + 1
        flt = getattr(self, "profile_filter", None)
        if prof is None or (flt is not None and idx not in flt):
            return None
        return prof

    def _head_split(self, x, table, bias):
        """The head convolution on the split-half kernel: max |x| per sequence into a range slot (v2ce_absmax_batch), then
        v2ce_conv3d_head_f16x2 -- planar network input in, channels-last-16 activations out."""
        B, T, _, H, W = x.shape
        Wp = self._pitch(W)
        y = torch.empty((B, T, 2, H, Wp, 16), dtype=torch.float32, device=x.device)
        y.lw, y.c16 = W, True
        per_b = self._prep["absmax"].dim() == 3
        ax = self._prep["absmax"][self._slot]
        ay = y.absmax = self._prep["absmax"][self._slot + 1]
        self._slot += 2
        st = hip.stream_ptr(x.device)
        n = T * 2 * H * W
        if per_b:
            hip.check(hip.lib().v2ce_absmax_batch(x.data_ptr(), B, n, ax.data_ptr(), 2, st), "v2ce_absmax_batch")
        else:
            hip.check(hip.lib().v2ce_absmax_batch(x.data_ptr(), 1, B * n, ax.data_ptr(), 2, st), "v2ce_absmax_batch")
        d = hip.ConvDesc(B=B, T=T, C0=2, H0=H, W0=W, C1=0, Hin=H, Win=W, Cout=32, Hout=H, Wout=W, ksize=3, stride_hw=1,
                         act=hip.ACT_LEAKY, tile_t=0, tile_h=0, tile_w=0, precision=hip.PRECISION_F16X2, W0_pitch=W, Win_pitch=W,
                         Wout_pitch=Wp, layout=hip.LAYOUT_C16, absmax_batch_stride=2 if per_b else 0)
        prof = self._prof_list()
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(hip.lib().v2ce_conv3d_head_f16x2(ctypes.byref(d), x.data_ptr(), table.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                   ax.data_ptr(), ay.data_ptr(), st), "v2ce_conv3d_head_f16x2")
        if prof is not None:
            e1.record()
            flops = 2.0 * B * T * H * W * 32 * 54
            prof.append(("conv3d_head_f16x2_kernel", flops, e0, e1, flops))
        return y

    def _sn_weight(self, inner: _SNConvInner, out):
        """spectral_norm.py:19-31: one power iteration (u, v updated in place), W_bar/sigma packed."""
        P = self._prep
        rows, cols = inner.weight_bar.shape[0], inner.weight_bar[0].numel()
        hip.check(hip.lib().v2ce_sn_power_iter(inner.weight_u.data_ptr(), inner.weight_v.data_ptr(),
                                               inner.weight_bar.data_ptr(), rows, cols,
                                               P["sigma"].data_ptr(), P["sn_ws"].data_ptr(),
                                               P["sn_ws"].numel(),
                                               hip.stream_ptr(inner.weight_bar.device)),
                  "v2ce_sn_power_iter")
        return self._pack(inner.weight_bar, P["sigma"], out, split=out.dtype == torch.float16)

    def _block(self, blk: _ResidualBlock3D, d, x0, x1=None, up_to=None, pred=None):
        """submodules.py:249-264: relu(bn2(conv2(relu(bn1(conv1 x)))) + bn_d(conv_d x))."""
        s = blk.stride_hw
        if blk.sn:
            self._await_sn()
        w1 = d["conv1_w"]
        track = self.precision == "f16x2"      # the block output may feed a split-half conv
        # (a spectral-norm layer whose planes were packed once carries 1 / sigma in its epilogue scale: _prepare, "sn_once")
        bn1, bn2 = d.get("bn1_eff", d["bn1"]), d.get("bn2_eff", d["bn2"])
        if (pred is not None and d.get("fold_lo") is not None and d.get("fold") is not None and blk.cout == 32 and x1 is not None
                and up_to is not None and self._half_up(x0, up_to) and self._up2_ok(x0, x1, w1, up_to)):
            # the last decoder block, its shortcut split by source (_dec_last_split): plain phase-folded conv1, the upsampled channels'
            # share of the shortcut at the source's resolution, conv2 + the skip channels' tail + pred in one launch
            t = self._conv(x0, x1, w1, *bn1, blk.cout, 3, s, hip.ACT_RELU, up_to=up_to, split=True)
            fw, fscale, fshift = d["fold"]
            r0 = self._conv(x0, None, d["fold_lo"], fscale, d["zero_shift"], blk.cout, 1, 1, hip.ACT_NONE, split=True, algo_hw=up_to)
            return self._conv(t, None, d["conv2_w"], bn2[0], fshift, blk.cout, 3, 1, hip.ACT_RELU, split=True, pred=pred,
                              tail=(x1, None, None, 1, d["fold_skip"]), residual=r0, residual_up=True)
        if self._fuse_shortcut(blk):
            t, res = self._conv(x0, x1, w1, *bn1, blk.cout, 3, s, hip.ACT_RELU, up_to=up_to, split=True,
                                sc=(d["down_w"], *d["down_bn"]))
        elif d.get("fold") is not None and pred is None:
            # the shortcut rides in conv2's K loop (v2ce_conv3d_fwd_tail): relu(s2 (W2 * t + Wd' * x) + shift2 + shift_d)
            if d.get("conv1_skip_w") is not None and x1 is not None and self._up2_ok(x0, x1, w1, up_to):
                part = self._conv_up_part(x0, x1, w1, *bn1, blk.cout, up_to)
                t = self._conv(x1, None, d["conv1_skip_w"], bn1[0], d["zero_shift"], blk.cout, 3, 1, hip.ACT_RELU, residual=part,
                               split=True)
            else:
                t = self._conv(x0, x1, w1, *bn1, blk.cout, 3, s, hip.ACT_RELU, up_to=up_to, split=True)
            fw, fscale, fshift = d["fold"]
            # conv2's scale: bn2's, over sigma when the planes are W_bar's (the folded shortcut then arrives as Wd' sigma: _sn_all);
            # the low-resolution share below is added BEHIND the scale and keeps bn2's own
            cscale = bn2[0]
            if d.get("fold_lo") is not None and x1 is not None and up_to is not None and self._half_up(x0, up_to):
                r0 = self._conv(x0, None, d["fold_lo"], fscale, d["zero_shift"], blk.cout, 1, 1, hip.ACT_NONE, split=True, algo_hw=up_to)
                return self._conv(t, None, d["conv2_w"], cscale, fshift, blk.cout, 3, 1, hip.ACT_RELU, split=True, track=track,
                                  tail=(x1, None, None, 1, d["fold_skip"]), residual=r0, residual_up=True)
            return self._conv(t, None, d["conv2_w"], cscale, fshift, blk.cout, 3, 1, hip.ACT_RELU, split=True, track=track,
                              tail=(x0, x1, up_to, s, fw))
        else:
            t = self._conv(x0, x1, w1, *bn1, blk.cout, 3, s, hip.ACT_RELU, up_to=up_to,
                           split=self._split(blk.cin, blk.cout))
            res = self._conv(x0, x1, d["down_w"], *d["down_bn"], blk.cout, 1, s, hip.ACT_NONE, up_to=up_to,
                             split=self._split(blk.cin, blk.cout, 1, s))
        w2 = d["conv2_w"]
        return self._conv(t, None, w2, *bn2, blk.cout, 3, 1, hip.ACT_RELU, residual=res,
                          split=self._split(blk.cout, blk.cout), track=track and pred is None, pred=pred)

    def _launch_sn(self):
        """One power iteration + re-pack for all 12 spectral-norm layers (the trajectory does not depend
        on the input): one batched call in line; with V2CE_SN_SIDE_STREAM=1 on a side stream that `_await_sn`
        joins behind the head conv (experiment, slower)."""
        dev = self.UNet.head.conv3d.weight.device
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_sn_stream", None) is None or self._sn_stream.device != dev:
            self._sn_stream = torch.cuda.Stream(device=dev)
        side = self._sn_stream
        if not os.environ.get("V2CE_SN_SIDE_STREAM"):
            # default: in line, in front of the head convolution (0.29 ms of kernels; measured cost in the forward
            # pass 0.59 ms -- the re-packed weights and the three passes over W also cool L2 / MALL for the
            # convolutions).  The side stream measured 0.64-0.75 ms overlapped with the head only and 0.66-0.94
            # overlapped with the encoder: see _forward.
            self._sn_all()
            self._sn_event = None
            return
        side.wait_stream(main)                   # the previous call's convs are done with the packed weights
        with torch.cuda.stream(side):
            self._sn_all()
            self._sn_event = torch.cuda.Event()
            self._sn_event.record(side)

    def _await_sn(self):
        ev = getattr(self, "_sn_event", None)
        if ev is not None:
            torch.cuda.current_stream(self.UNet.head.conv3d.weight.device).wait_event(ev)
            self._sn_event = None

    @torch.no_grad()
    def advance_spectral_norm(self):
        """Apply the power iteration of one forward call to all 12 SN layers without running the
        convolutions (the u/v trajectory is input independent): used to fast-forward a replica to
        the global call index it emulates when calls are sharded over GPUs."""
        with torch.cuda.device(self.UNet.head.conv3d.weight.device):
            self._advance_spectral_norm()

    def _sn_all(self):
        """One power iteration + re-pack for all spectral-norm layers on the current stream."""
        batch = self._prep["sn_batch"]
        if batch is not None:
            arr, n, ws = batch
            dev = ws.device
            hip.check(hip.lib().v2ce_sn_update_batch(arr, n, ws.data_ptr(), ws.numel(), hip.stream_ptr(dev)),
                      "v2ce_sn_update_batch")
            once = self._prep.get("sn_once")
            if once is not None and once["tails"]:            # the folded shortcuts Wd' sigma (pack-only layers: one launch)
                tarr, tn, tws = once["tail_batch"]
                hip.check(hip.lib().v2ce_sn_update_batch(tarr, tn, tws.data_ptr(), tws.numel(), hip.stream_ptr(dev)),
                          "v2ce_sn_update_batch")
            return
        once = self._prep.get("sn_once")
        k = 0
        for name, blocks in (("res", self.UNet.resblocks), ("dec", self.UNet.decoders)):
            for i, blk in enumerate(blocks):
                d = self._prep[f"{name}{i}"]
                if once is not None:
                    # the per-layer form of the pack-once scheme (tests): power iteration, then scale / sigma and 1 / sigma by
                    # the same IEEE divisions the batched kernels make
                    for cn, bn in (("conv1", "bn1"), ("conv2", "bn2")):
                        inner = getattr(blk, cn).module
                        rows, cols = inner.weight_bar.shape[0], inner.weight_bar[0].numel()
                        P = self._prep
                        hip.check(hip.lib().v2ce_sn_power_iter(inner.weight_u.data_ptr(), inner.weight_v.data_ptr(), inner.weight_bar.data_ptr(),
                                                               rows, cols, P["sigma"].data_ptr(), P["sn_ws"].data_ptr(), P["sn_ws"].numel(),
                                                               hip.stream_ptr(inner.weight_bar.device)), "v2ce_sn_power_iter")
                        torch.div(d[bn][0], P["sigma"], out=d[bn + "_eff"][0])
                        torch.div(once["one"], P["sigma"], out=once["inv_sigma"][k:k + 1])
                        k += 1
                    continue
                self._sn_weight(blk.conv1.module, d["conv1_w"])
                if d.get("conv1_skip_w") is not None:          # (P["sigma"] still holds conv1's)
                    self._pack(blk.conv1.module.weight_bar, self._prep["sigma"], d["conv1_skip_w"], split=True)
                self._sn_weight(blk.conv2.module, d["conv2_w"])
        if once is not None:
            for w32, buf, kk, _ in once["tails"]:
                self._pack(w32.view(*w32.shape[:2], 1, 1, 1), once["inv_sigma"][kk:kk + 1], buf, split=True)

    def _advance_spectral_norm(self):
        if self._prep is None:
            self._prepare()
        self._sn_all()
        self.calls += 1

    # ---- forward ------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, return_intermediates: bool = False):
        """x [B,L,2,H,W] f32 on the device -> [B,L,20,H,W] (contiguous; the reference returns the
        same values as a permuted view, v2ce_3d.py:29).  Like the reference, every call advances
        the spectral-norm u/v of the 12 SN layers by one power iteration."""
        if x.dim() != 5 or x.shape[2] != self.in_channels:
            raise ValueError(f"expected x of shape [B,L,{self.in_channels},H,W], got {tuple(x.shape)}")
        x = hip.require_device_f32(x, "x")
        # every launch below goes to the current stream of x's device through the C ABI: make that
        # device current for the whole call (the ABI has no device argument, like a HIP stream call)
        with torch.cuda.device(x.device):
            if self.precision != "f16x2" or self.guard != "call":
                return self._forward(x, return_intermediates)
            snap = self._sn_snapshot_fast()
            out = self._forward(x, return_intermediates)
            worst = float(self._prep["absmax"][:, :, 1].max().item())        # this call's bound (synchronises)
            if worst <= self.RANGE_GUARD_LIMIT:
                return out
            import logging
            logging.getLogger("V2CE").warning(
                f"split-half range guard: bound {worst:.3e} > {self.RANGE_GUARD_LIMIT:.1e}; repeating this call on the "
                "exact-f32 kernels")
            del out
            self._sn_restore_fast(snap)
            self.guard_reruns += 1
            # (switching precision rebuilds the derived constants on the way in and out: a one-off for a checkpoint
            # that needs it on every call -- construct it with precision='f32' instead)
            with self.exact_f32():
                return self._forward(x, return_intermediates)

    def _forward(self, x, return_intermediates):
        if self._prep is None:
            self._prepare()
        P, U = self._prep, self.UNet
        self._slot = 0
        self._launch_idx = 0
        if self.precision == "f16x2":
            if P["absmax"].shape[1] != x.shape[0]:
                P["absmax"] = torch.zeros((64, x.shape[0], 2), dtype=torch.float32, device=x.device)
            else:
                P["absmax"].zero_()
        self._launch_sn()
        inter = OrderedDict()
        if P["head_split"] is not None:
            h = self._head_split(x, *P["head_split"])                               # unet_2layer.py:341
        else:
            h = self._conv(x, None, *P["head"], BASE, 3, 1, hip.ACT_LEAKY,           # unet_2layer.py:341
                           track=self.precision == "f16x2")
        inter["head"] = self.to_planar(h) if return_intermediates else None
        # The spectral-norm stream overlaps the head convolution only (both are many small workgroups; the head
        # is bound by its output stream, the power iterations by reading W).  The persistent residual-block
        # kernels need a whole CU per workgroup and walk their tiles statically: side-stream workgroups that
        # sit on a CU when such a kernel starts delay that CU's whole share (measured: 0.29 ms of spectral-norm
        # work cost 0.66 ms when it overlapped the encoder).
        self._await_sn()
        skips = []
        for i, blk in enumerate(U.encoders):                                     # :345-347
            skips.append(h)
            h = self._block(blk, P[f"enc{i}"], h)
            inter[f"enc{i}"] = self.to_planar(h) if return_intermediates else None
        for i, blk in enumerate(U.resblocks):                                    # :349-350
            h = self._block(blk, P[f"res{i}"], h)
            inter[f"res{i}"] = self.to_planar(h) if return_intermediates else None
        fuse = P["pred_fused"] if not return_intermediates else None
        for i, (blk, skip) in enumerate(zip(U.decoders, reversed(skips))):       # :357-365
            last = i == len(U.decoders) - 1
            h = self._block(blk, P[f"dec{i}"], h, skip, up_to=(skip.shape[3], getattr(skip, "lw", skip.shape[4])),
                            pred=fuse if last else None)
            inter[f"dec{i}"] = self.to_planar(h) if return_intermediates else None
        if fuse is not None:
            out = h                                                               # pred rode on dec3.conv2
        else:
            out = self._conv(h, None, *P["pred"], self.out_channels, 1, 1, hip.ACT_RELU, dense_out=True)   # :374
        self.calls += 1
        if self.precision == "f16x2":
            torch.maximum(P["guard"], P["absmax"][:, :, 1].max().reshape(1), out=P["guard"])
        if return_intermediates:
            return out, inter
        return out

    # ---- range guard of the split-half path (include/v2ce_hip.h "Range guard") ---------------
    RANGE_GUARD_LIMIT = 2.5e-6

    def range_guard_value(self, reset: bool = True) -> float:
        """Largest per-launch guard bound E reported by the split-half convolutions since the last reset
        (synchronises).  E <= RANGE_GUARD_LIMIT: every layer's output is provably within that distance
        of the exact-f32 kernels'.  Larger: the activations' dynamic range is beyond what one scale per
        tensor covers -- repeat on ``exact_f32()`` (glue.run_guarded does)."""
        if self._prep is None or self.precision != "f16x2":
            return 0.0
        v = float(self._prep["guard"].item())
        if reset:
            self._prep["guard"].zero_()
        return v

    def _sn_params(self):
        if getattr(self, "_sn_plist", None) is None:
            self._sn_plist = [p for n, p in self.named_parameters() if n.endswith(("weight_u", "weight_v"))]
        return self._sn_plist

    def _sn_snapshot_fast(self):
        """u / v of the 12 SN layers into persistent buffers with one multi-tensor copy (per-call guard)."""
        ps = self._sn_params()
        bufs = getattr(self, "_sn_bufs", None)
        if bufs is None or bufs[0].device != ps[0].device:
            bufs = self._sn_bufs = [torch.empty_like(p) for p in ps]
        torch._foreach_copy_(bufs, [p.data for p in ps])
        return bufs, self.calls

    def _sn_restore_fast(self, snap):
        bufs, calls = snap
        torch._foreach_copy_([p.data for p in self._sn_params()], bufs)
        self.calls = calls

    def sn_snapshot(self):
        """The state a forward call mutates (spectral-norm u / v of the 12 SN layers, call counter)."""
        st = [(p, p.detach().clone()) for n, p in self.named_parameters() if n.endswith(("weight_u", "weight_v"))]
        return st, self.calls

    def sn_restore(self, snap):
        st, calls = snap
        with torch.no_grad():
            for p, v in st:
                p.copy_(v)
        self.calls = calls

    def exact_f32(self):
        """Context manager: run on the exact-f32 kernels (precision 'f32'), then switch back."""
        model = self

        class _Exact:
            def __enter__(self):
                self.prev = model.precision
                model.precision, model._prep = "f32", None

            def __exit__(self, *exc):
                model.precision, model._prep = self.prev, None
                return False
        return _Exact()
