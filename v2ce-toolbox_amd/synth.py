"""Deterministic synthetic inputs: weights with the reference state_dict layout, and frames.

The pretrained ``weights/v2ce_3d.pt`` is a Google-Drive download that is absent from the reference
checkout (``/root/reference/readme.md:13``, ``weights/.gitkeep``), so every test, golden vector and
benchmark uses this generator.  It emits exactly the 218 keys / shapes / dtypes, in the order, of
``V2ce3d().state_dict()`` (``scripts/v2ce_3d.py:13-24``, ``scripts/unet_2layer.py:203-318``,
``scripts/submodules.py:85-124,216-264``, ``scripts/spectral_norm.py:43-59``).

Numbers come from ``numpy.random.Generator(Philox(seed))`` so the build container (oracle, goldens)
and the GPU box produce identical tensors.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

BASE = 32
NUM_ENCODERS = 4
NUM_RESBLOCKS = 2
IN_CH, OUT_CH = 2, 20


def layer_plan():
    """[(prefix, kind, cin, cout)] in state_dict order; kind in {'enc','res','dec'}."""
    plan = []
    for i in range(NUM_ENCODERS):
        plan.append((f"UNet.encoders.{i}", "enc", BASE * 2 ** i, BASE * 2 ** (i + 1)))
    cmax = BASE * 2 ** NUM_ENCODERS
    for i in range(NUM_RESBLOCKS):
        plan.append((f"UNet.resblocks.{i}", "res", cmax, cmax))
    for i in range(NUM_ENCODERS):
        cout_enc = BASE * 2 ** (NUM_ENCODERS - i)          # 512, 256, 128, 64
        plan.append((f"UNet.decoders.{i}", "dec", int(1.5 * cout_enc), cout_enc // 2))
    return plan


_TAILS = {"family": "normal"}      # set by make_state_dict for the duration of a call


def _normal(rng, shape, std):
    if _TAILS["family"] == "student":
        # Student-t with 4 degrees of freedom, rescaled to the requested standard deviation (variance of t_4 = 2): the
        # largest of a million weights sits ~30 sigma out instead of ~5 -- the tail a trained checkpoint may have
        t = rng.standard_t(4, shape).astype(np.float32) / np.float32(2.0 ** 0.5)
        return torch.from_numpy(t * np.float32(std))
    return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(std)))


def _bn(rng, sd, prefix, c, gamma):
    sd[prefix + ".weight"] = torch.from_numpy(
        (gamma * (1.0 + 0.1 * rng.standard_normal(c))).astype(np.float32))
    sd[prefix + ".bias"] = torch.from_numpy((0.05 * rng.standard_normal(c)).astype(np.float32))
    sd[prefix + ".running_mean"] = torch.from_numpy(
        (0.05 * rng.standard_normal(c)).astype(np.float32))
    sd[prefix + ".running_var"] = torch.from_numpy(
        (1.0 + 0.2 * rng.random(c)).astype(np.float32))
    if _TAILS["family"] == "student":
        # a tenth of the channels with running_var log-uniform in [1e-3, 1] and gamma moved with sqrt(var) x U[0.5, 2): the
        # folded scale gamma / sqrt(var + eps) spreads over a factor of four instead of 1.2, the statistics over three decades
        pick = rng.random(c) < 0.1
        var = np.exp(rng.uniform(np.log(1e-3), 0.0, c))
        spread = rng.uniform(0.5, 2.0, c)
        rv = sd[prefix + ".running_var"].numpy().copy()
        w = sd[prefix + ".weight"].numpy().copy()
        w[pick] = (w[pick] * np.sqrt(var[pick] / rv[pick]) * spread[pick]).astype(np.float32)
        rv[pick] = var[pick].astype(np.float32)
        sd[prefix + ".running_var"], sd[prefix + ".weight"] = torch.from_numpy(rv), torch.from_numpy(w)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)


def _unit(rng, n):
    v = rng.standard_normal(n)
    return torch.from_numpy((v / np.linalg.norm(v)).astype(np.float32))


def make_state_dict(seed: int = 0, out_gain: float = 1.0, tails: str = "normal") -> "OrderedDict[str, torch.Tensor]":
    """Synthetic ``V2ce3d`` weights (reference key layout).

    Scales are chosen so that activations neither die nor explode through the 22 conv layers and
    the final ReLU output has a DVS-like distribution (mostly < 1, a tail above 1) so LDATI is
    exercised in all three regimes (no event / single event / multiple events per voxel).
    tails: "normal" (Gaussian weights, BatchNorm variances in [1, 1.2]) or "student" (Student-t_4 weights and a tenth of
    the BatchNorm channels with variances down to 1e-3: a heavier-tailed stand-in for a trained checkpoint).
    """
    if tails not in ("normal", "student"):
        raise ValueError(tails)
    _TAILS["family"] = tails
    try:
        return _make_state_dict(seed, out_gain)
    finally:
        _TAILS["family"] = "normal"


def converge_spectral_norm(sd, iterations: int = 50, separate: float = 0.0):
    """u / v of every spectral-norm layer after `iterations` power iterations (spectral_norm.py:19-31) -- where a trained
    checkpoint's are (the generator's are random unit vectors): in place, f64 arithmetic, returns sd.
    separate > 0: first add separate * sigma_1 * u_1 v_1^T to every weight_bar.  The leading singular values of an i.i.d.
    Gaussian matrix are nearly degenerate (the power iteration needs ~1e4 steps there); a trained layer has a dominant
    direction, and with a gap of 1 + separate fifty iterations reach the fixed point to rounding."""
    for k in [k for k in sd if k.endswith(".weight_bar")]:
        w = sd[k].double().reshape(sd[k].shape[0], -1)
        if separate > 0:
            U_, S_, Vh_ = torch.linalg.svd(w, full_matrices=False)
            w = w + separate * S_[0] * torch.outer(U_[:, 0], Vh_[0])
            sd[k] = w.float().reshape(sd[k].shape).contiguous()
            w = sd[k].double().reshape(sd[k].shape[0], -1)
        u, v = sd[k[:-4] + "_u"].double(), sd[k[:-4] + "_v"].double()
        for _ in range(iterations):
            v = w.t() @ u
            v = v / (v.norm() + 1e-12)
            u = w @ v
            u = u / (u.norm() + 1e-12)
        sd[k[:-4] + "_u"], sd[k[:-4] + "_v"] = u.float(), v.float()
    return sd


def _make_state_dict(seed, out_gain):
    rng = np.random.Generator(np.random.Philox(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd["UNet.head.conv3d.weight"] = _normal(rng, (BASE, IN_CH, 3, 3, 3), (2.0 / (IN_CH * 27)) ** 0.5)
    sd["UNet.head.conv3d.bias"] = _normal(rng, (BASE,), 0.05)
    for prefix, kind, cin, cout in layer_plan():
        sn = kind != "enc"
        k1, k2 = cin * 27, cout * 27
        if sn:
            # spectral norm divides by sigma ~ std*(sqrt(cout)+sqrt(K)); BN gamma restores the scale
            sd[prefix + ".conv1.module.weight_u"] = _unit(rng, cout)
            sd[prefix + ".conv1.module.weight_v"] = _unit(rng, k1)
            sd[prefix + ".conv1.module.weight_bar"] = _normal(rng, (cout, cin, 3, 3, 3), (2.0 / k1) ** 0.5)
            g1 = 1.0 + (cout / k1) ** 0.5
            g2 = 1.0 + (cout / k2) ** 0.5
        else:
            sd[prefix + ".conv1.weight"] = _normal(rng, (cout, cin, 3, 3, 3), (2.0 / k1) ** 0.5)
            g1 = g2 = 1.0
        _bn(rng, sd, prefix + ".bn1", cout, 0.8 * g1)
        _bn(rng, sd, prefix + ".bn2", cout, 0.6 * g2)
        if sn:
            sd[prefix + ".conv2.module.weight_u"] = _unit(rng, cout)
            sd[prefix + ".conv2.module.weight_v"] = _unit(rng, k2)
            sd[prefix + ".conv2.module.weight_bar"] = _normal(rng, (cout, cout, 3, 3, 3), (2.0 / k2) ** 0.5)
        else:
            sd[prefix + ".conv2.weight"] = _normal(rng, (cout, cout, 3, 3, 3), (2.0 / k2) ** 0.5)
        sd[prefix + ".downsample.0.weight"] = _normal(rng, (cout, cin, 1, 1, 1), (1.0 / cin) ** 0.5)
        sd[prefix + ".downsample.0.bias"] = _normal(rng, (cout,), 0.05)
        _bn(rng, sd, prefix + ".downsample.1", cout, 0.6)
    sd["UNet.pred.conv3d.weight"] = _normal(rng, (OUT_CH, BASE, 1, 1, 1), out_gain * 0.9 / BASE ** 0.5)
    sd["UNet.pred.conv3d.bias"] = _normal(rng, (OUT_CH,), 0.25) - 0.35
    return sd


def synthetic_frames(num_frames: int, height: int = 260, width: int = 346, seed: int = 1,
                     pattern: str = "drift") -> np.ndarray:
    """uint8 grayscale frames [N,H,W].

    ``drift``: two drifting sinusoids + 2 % salt noise (moving-scene proxy).
    ``checker``: high-contrast checkerboard that flips every frame (high event-rate stress).
    ``noise``: i.i.d. uniform uint8, the reference authors' own dummy input
    (``train/scripts/tools/dummy_data_gen.py``).
    """
    rng = np.random.Generator(np.random.Philox(seed))
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    out = np.empty((num_frames, height, width), np.uint8)
    for i in range(num_frames):
        if pattern == "drift":
            ph = 0.35 * i
            img = 0.5 + 0.25 * np.sin(0.045 * xx + 0.02 * yy + ph) + 0.2 * np.sin(0.03 * yy - 0.6 * ph)
            img = np.clip(img, 0.0, 1.0) * 255.0
            salt = rng.random((height, width)) < 0.02
            img = np.where(salt, rng.integers(0, 256, (height, width)), img)
        elif pattern == "checker":
            cb = ((xx.astype(np.int32) // 8 + yy.astype(np.int32) // 8 + i) % 2).astype(np.float32)
            img = cb * 255.0
        elif pattern == "noise":
            img = rng.integers(0, 256, (height, width))
        else:
            raise ValueError(pattern)
        out[i] = np.asarray(img).astype(np.uint8)
    return out


def synthetic_voxels(num_pairs: int, height: int = 260, width: int = 346, seed: int = 0,
                     regime: str = "sparse") -> np.ndarray:
    """float32 voxel grids [L,2,10,H,W] for LDATI-only runs.

    ``sparse``: relu(0.8*randn)  (~6.8 events/pixel/frame);  ``frac``: U[0,0.99) (all-fractional);
    ``stress``: 6*U[0,1) (~59 events/pixel/frame).
    """
    rng = np.random.Generator(np.random.Philox(seed))
    shape = (num_pairs, 2, 10, height, width)
    if regime == "sparse":
        v = np.maximum(0.8 * rng.standard_normal(shape, dtype=np.float32), 0.0)
    elif regime == "frac":
        v = 0.99 * rng.random(shape, dtype=np.float32)
    elif regime == "stress":
        v = 6.0 * rng.random(shape, dtype=np.float32)
    else:
        raise ValueError(regime)
    return v.astype(np.float32)
