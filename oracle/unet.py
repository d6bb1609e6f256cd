"""Stage-1 oracle: functional torch-fp32 (CPU) restatement of ``V2ce3d``.  TEST INFRASTRUCTURE.

A floating-point path, so this is the "plain PyTorch fp32 reference" the HIP kernels are compared
against (tolerance 1e-5 abs + 1e-5 rel, BASELINE.json north_star).  It restates

* ``scripts/v2ce_3d.py:26-30``          -- permutes around the UNet, take the last prediction
* ``scripts/unet_2layer.py:335-379``    -- head, 4 encoders, 2 res-blocks, 4 x (nearest upsample of
  H,W to the skip's size, concat [upsampled, skip], decoder), 1x1x1 prediction + ReLU
* ``scripts/submodules.py:115-124``     -- ConvLayer3D: conv -> (no norm) -> LeakyReLU(0.01) / relu
* ``scripts/submodules.py:249-264``     -- ResidualBlock3D: relu(bn2(conv2(relu(bn1(conv1 x)))) +
  bn_d(conv_d(x))), BatchNorm3d in eval mode (running stats, eps 1e-5)
* ``scripts/spectral_norm.py:19-31``    -- one power iteration per forward, mutating u and v

as free functions over a state_dict with the reference key layout; no nn.Module of the reference
is used.  Pinned by tests/golden/unet_*.npz, which hold outputs of the reference itself.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)                     # spectral_norm.py:5-6


def sn_step(sd, prefix):
    """One power iteration; updates sd[prefix.weight_u/_v] in place and returns W_bar / sigma
    (spectral_norm.py:19-31)."""
    u, v, w = sd[prefix + ".weight_u"], sd[prefix + ".weight_v"], sd[prefix + ".weight_bar"]
    height = w.shape[0]
    wm = w.reshape(height, -1)
    v = _l2normalize(torch.mv(wm.t(), u))
    u = _l2normalize(torch.mv(wm, v))
    sigma = u.dot(wm.mv(v))
    sd[prefix + ".weight_u"], sd[prefix + ".weight_v"] = u, v
    return w / sigma.expand_as(w)


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def residual_block(sd, prefix, x, stride, sn):
    """submodules.py:249-264 (shortcut conv always present: submodules.py:220,247)."""
    if sn:
        w1 = sn_step(sd, prefix + ".conv1.module")
    else:
        w1 = sd[prefix + ".conv1.weight"]
    out = F.conv3d(x, w1, None, stride, 1)
    out = torch.relu(_bn(sd, prefix + ".bn1", out))
    if sn:
        w2 = sn_step(sd, prefix + ".conv2.module")
    else:
        w2 = sd[prefix + ".conv2.weight"]
    out = _bn(sd, prefix + ".bn2", F.conv3d(out, w2, None, 1, 1))
    res = F.conv3d(x, sd[prefix + ".downsample.0.weight"], sd[prefix + ".downsample.0.bias"],
                   stride, 0)
    res = _bn(sd, prefix + ".downsample.1", res)
    return torch.relu(out + res)


def upsample_nearest_hw(x, size):
    """unet_2layer.py:358-362: nearest interpolation of (H,W) only; depth untouched."""
    B, C, L, H, W = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(B * L, C, H, W)
    y = F.interpolate(y, size=size, mode="nearest")
    return y.reshape(B, L, C, size[0], size[1]).permute(0, 2, 1, 3, 4)


@torch.no_grad()
def forward(sd: "OrderedDict[str, torch.Tensor]", x: torch.Tensor, return_intermediates=False):
    """x [B,L,2,H,W] f32 -> [B,L,20,H,W] f32; mutates the SN u/v entries of ``sd`` exactly like one
    reference ``model(x)`` call."""
    inter = OrderedDict()
    h = x.permute(0, 2, 1, 3, 4)                                        # v2ce_3d.py:27
    h = F.conv3d(h, sd["UNet.head.conv3d.weight"], sd["UNet.head.conv3d.bias"], 1, 1)
    h = F.leaky_relu(h, 0.01)                                           # submodules.py:101-103
    inter["head"] = h
    skips = []
    for i in range(4):                                                  # unet_2layer.py:345-347
        skips.append(h)
        h = residual_block(sd, f"UNet.encoders.{i}", h, (1, 2, 2), sn=False)
        inter[f"enc{i}"] = h
    for i in range(2):                                                  # unet_2layer.py:349-350
        h = residual_block(sd, f"UNet.resblocks.{i}", h, (1, 1, 1), sn=True)
        inter[f"res{i}"] = h
    for i, skip in enumerate(reversed(skips)):                          # unet_2layer.py:357-365
        h = upsample_nearest_hw(h, (skip.shape[3], skip.shape[4]))
        h = torch.cat([h, skip], dim=1)                                 # upsampled first, skip second
        h = residual_block(sd, f"UNet.decoders.{i}", h, (1, 1, 1), sn=True)
        inter[f"dec{i}"] = h
    h = F.conv3d(h, sd["UNet.pred.conv3d.weight"], sd["UNet.pred.conv3d.bias"], 1, 0)
    h = torch.relu(h)                                                   # unet_2layer.py:290-298
    out = h.permute(0, 2, 1, 3, 4)                                      # v2ce_3d.py:29
    if return_intermediates:
        return out, inter
    return out


def clone_state(sd):
    return OrderedDict((k, v.clone()) for k, v in sd.items())
