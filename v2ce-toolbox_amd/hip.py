"""ctypes binding of ``csrc/libv2ce_hip.so`` (C ABI: ``include/v2ce_hip.h``).

PyTorch is used only for device memory and streams: every call passes ``tensor.data_ptr()`` and
the current HIP stream handle.  There is NO fallback: if the library is missing or cannot be
loaded, ``lib()`` raises, and so does every op built on it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("V2CE_HIP_LIB", os.path.join(CSRC, "libv2ce_hip.so"))   # override: kernel A/B builds

RNG_REPLAY, RNG_PHILOX = 0, 1
STRATEGY_SLOPE, STRATEGY_NONE, STRATEGY_RANDOM = 0, 1, 2
POOL_NONE, POOL_AVG, POOL_WEIGHTED = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
PRECISION_F32, PRECISION_F16X2 = 0, 1

EXPORTS = [
    "v2ce_version", "v2ce_last_error", "v2ce_ldati_count", "v2ce_ldati_rank_mode", "v2ce_ldati_tile_ws_bytes", "v2ce_ldati_status", "v2ce_ldati_plan_info",
    "v2ce_ldati_selfcheck", "v2ce_ldati_fused_ws_bytes", "v2ce_ldati_count_fused", "v2ce_ldati_emit_fused", "v2ce_ldati_lds_bytes", "v2ce_ldati_workspace_bytes", "v2ce_ldati_emit", "v2ce_events_pack", "v2ce_events_unpack",
    "v2ce_conv3d_fwd",
    "v2ce_conv3d_variant", "v2ce_conv3d_variant_fused", "v2ce_pack_weights_f16x2", "v2ce_pack_weights_f16x2_bytes",
    "v2ce_pack_weights", "v2ce_sn_workspace_bytes", "v2ce_sn_power_iter", "v2ce_sn_batch_workspace_bytes", "v2ce_sn_update_batch",
    "v2ce_preprocess_pairs", "v2ce_preprocess_pairs_resize",
    "v2ce_voxelize_events", "v2ce_conv3d_fwd_pred", "v2ce_conv3d_fwd_sc", "v2ce_conv3d_fwd_tail", "v2ce_conv3d_fwd_tail_pred", "v2ce_pack_pred_weights_f16x2", "v2ce_pack_pred_weights_f16x2_bytes",
    "v2ce_sampler_count", "v2ce_sampler_workspace_bytes", "v2ce_sampler_emit", "v2ce_sampler_pool",
    "v2ce_conv3d_fwd_up2", "v2ce_pack_weights_f16x2_up", "v2ce_pack_weights_f16x2_up_bytes", "v2ce_conv3d_up2_variant",
    "v2ce_conv3d_fwd_wt", "v2ce_conv3d_fwd_wt_tail", "v2ce_pack_weights_f16x2_wt", "v2ce_pack_weights_f16x2_wt_slice", "v2ce_conv3d_fwd_up2_part", "v2ce_pack_weights_f16x2_wt_bytes", "v2ce_conv3d_wt_variant",
    "v2ce_conv3d_head_f16x2", "v2ce_pack_head_weights_f16x2", "v2ce_pack_head_weights_f16x2_bytes", "v2ce_absmax_batch",
]


LAYOUT_PLANAR, LAYOUT_C16 = 0, 1


class ConvDesc(ctypes.Structure):
    """``v2ce_conv3d_desc`` (include/v2ce_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "T", "C0", "H0", "W0", "C1", "Hin", "Win", "Cout", "Hout", "Wout", "ksize",
        "stride_hw", "act", "tile_t", "tile_h", "tile_w", "precision", "W0_pitch", "Win_pitch", "Wout_pitch", "layout",
        "absmax_batch_stride")]


class SnLayer(ctypes.Structure):
    """``v2ce_sn_layer`` (include/v2ce_hip.h): one spectral-norm layer of v2ce_sn_update_batch."""
    _fields_ = [("w_bar", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("packed", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("k3", ctypes.c_int32), ("up_c0", ctypes.c_int32),
                ("wt", ctypes.c_int32), ("flags", ctypes.c_int32), ("packed_skip", ctypes.c_void_p),
                ("bn_scale", ctypes.c_void_p), ("scale_out", ctypes.c_void_p), ("inv_sigma_out", ctypes.c_void_p),
                ("sigma_src", ctypes.c_void_p), ("wmax", ctypes.c_float), ("reserved", ctypes.c_int32)]


SN_NO_PACK, SN_NO_ITERATE = 1, 2


class LdatiOptions(ctypes.Structure):
    """``v2ce_ldati_options`` (include/v2ce_hip.h): the keyword options of sample_voxel_statistical."""
    _fields_ = [(n, ctypes.c_int32) for n in ("strategy", "bidirectional", "pooling_type", "pooling_kernel_size")]


class SamplerOptions(ctypes.Structure):
    """``v2ce_sampler_options`` (include/v2ce_hip.h): the ablation samplers of SURVEY 8f4."""
    _fields_ = [("mode", ctypes.c_int32), ("rng_mode", ctypes.c_int32), ("fps", ctypes.c_double), ("t0", ctypes.c_double),
                ("seed", ctypes.c_uint64), ("frame_base", ctypes.c_int32), ("replay_M", ctypes.c_int32),
                ("u_int", ctypes.c_void_p), ("u_dec", ctypes.c_void_p), ("u_bern", ctypes.c_void_p), ("pooled", ctypes.c_void_p)]


SAMPLER_RANDOM, SAMPLER_EVEN, SAMPLER_PURE_SLOPE = 0, 1, 2


class V2ceHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4", "libv2ce_hip.so"])
    return SO_PATH


_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise V2ceHipError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (make -C {CSRC}).  There is no CPU fallback.")
    L = ctypes.CDLL(SO_PATH)
    vp, i32, i64, f64, u64, sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double,
                                  ctypes.c_uint64, ctypes.c_size_t)
    L.v2ce_version.restype = ctypes.c_char_p
    L.v2ce_last_error.restype = ctypes.c_char_p
    op = ctypes.POINTER(LdatiOptions)
    L.v2ce_ldati_count.argtypes = [vp, i32, i32, i32, op, vp, sz, vp, vp, vp]
    L.v2ce_ldati_tile_ws_bytes.argtypes = [i32, i32, i32]
    L.v2ce_ldati_tile_ws_bytes.restype = sz
    L.v2ce_ldati_lds_bytes.argtypes = [f64, f64]
    L.v2ce_ldati_lds_bytes.restype = sz
    L.v2ce_ldati_emit.argtypes = [vp, i32, i32, i32, f64, f64, op, i32, vp, i32, u64, i64, vp, vp,
                                  vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, sz, vp]
    L.v2ce_ldati_fused_ws_bytes.argtypes = [i32, i32, i32, f64, f64, op, i64, i64]
    L.v2ce_ldati_fused_ws_bytes.restype = sz
    L.v2ce_ldati_count_fused.argtypes = [vp, i32, i32, i32, f64, f64, op, i32, vp, i32, u64, i64, i64, i64, vp, sz, vp, sz, vp, vp, vp]
    L.v2ce_ldati_count_fused.restype = ctypes.c_int
    L.v2ce_ldati_emit_fused.argtypes = [vp, i32, i32, i32, f64, f64, op, i32, vp, i32, u64, i64, vp, vp,
                                        vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, sz, vp, sz, i64, i64, i64, vp]
    L.v2ce_ldati_emit_fused.restype = ctypes.c_int
    L.v2ce_ldati_workspace_bytes.argtypes = [i32, i32, i32, f64, f64, op, i64, i64, i64, i32]
    L.v2ce_ldati_workspace_bytes.restype = sz
    L.v2ce_ldati_status.argtypes = [vp, i32, i32, i32, f64, f64, op, i64, i64, i64, ctypes.POINTER(vp)]
    L.v2ce_ldati_status.restype = ctypes.c_int
    L.v2ce_ldati_plan_info.argtypes = [i32, i32, i32, f64, f64, op, i64, i64, i64, vp]
    L.v2ce_ldati_plan_info.restype = ctypes.c_int
    L.v2ce_ldati_selfcheck.argtypes = [f64, ctypes.POINTER(ctypes.c_int64)]
    L.v2ce_ldati_selfcheck.restype = ctypes.c_int
    L.v2ce_ldati_rank_mode.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    L.v2ce_ldati_rank_mode.restype = ctypes.c_int
    L.v2ce_events_pack.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.v2ce_events_unpack.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.v2ce_events_unpack.restype = ctypes.c_int
    L.v2ce_conv3d_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 13
    L.v2ce_pack_weights.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.v2ce_pack_weights_f16x2.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.v2ce_pack_weights_f16x2.restype = ctypes.c_int
    L.v2ce_pack_weights_f16x2_bytes.argtypes = [i32, i32, i32]
    L.v2ce_pack_weights_f16x2_bytes.restype = sz
    L.v2ce_conv3d_variant.argtypes = [ctypes.POINTER(ConvDesc), i32, ctypes.c_char_p, sz]
    L.v2ce_conv3d_variant_fused.argtypes = [ctypes.POINTER(ConvDesc), i32, i32, ctypes.c_char_p, sz]
    L.v2ce_conv3d_variant_fused.restype = ctypes.c_int
    L.v2ce_preprocess_pairs.argtypes = [vp, i32, i32, i32, ctypes.c_float, ctypes.c_float, vp, vp]
    L.v2ce_preprocess_pairs.restype = ctypes.c_int
    L.v2ce_preprocess_pairs_resize.argtypes = [vp, i32, i32, i32, i32, i32, ctypes.c_float, ctypes.c_float, vp, vp]
    L.v2ce_preprocess_pairs_resize.restype = ctypes.c_int
    L.v2ce_voxelize_events.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp]
    L.v2ce_voxelize_events.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_pred.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 14 + [i32, vp, vp]
    L.v2ce_conv3d_fwd_pred.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_sc.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 16
    L.v2ce_conv3d_fwd_sc.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_tail.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 11 + [ctypes.POINTER(ConvDesc)] + [vp] * 8
    L.v2ce_conv3d_fwd_tail.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_tail_pred.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 9 + [i32, vp, ctypes.POINTER(ConvDesc)] + [vp] * 8 + [i32, i32, vp]
    L.v2ce_conv3d_fwd_tail_pred.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_up2.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 14
    L.v2ce_conv3d_fwd_up2.restype = ctypes.c_int
    L.v2ce_pack_weights_f16x2_up.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.v2ce_pack_weights_f16x2_up.restype = ctypes.c_int
    L.v2ce_pack_weights_f16x2_up_bytes.argtypes = [i32, i32, i32]
    L.v2ce_pack_weights_f16x2_up_bytes.restype = sz
    L.v2ce_conv3d_up2_variant.argtypes = [ctypes.POINTER(ConvDesc), i32, ctypes.c_char_p, sz]
    L.v2ce_conv3d_up2_variant.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_wt.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 9
    L.v2ce_conv3d_fwd_wt.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_wt_tail.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 7 + [ctypes.POINTER(ConvDesc)] + [vp] * 8 + [i32, i32, vp]
    L.v2ce_conv3d_fwd_wt_tail.restype = ctypes.c_int
    L.v2ce_pack_weights_f16x2_wt_slice.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    L.v2ce_pack_weights_f16x2_wt_slice.restype = ctypes.c_int
    L.v2ce_conv3d_fwd_up2_part.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 9
    L.v2ce_conv3d_fwd_up2_part.restype = ctypes.c_int
    L.v2ce_pack_weights_f16x2_wt.argtypes = [vp, i32, i32, vp, vp, vp]
    L.v2ce_pack_weights_f16x2_wt.restype = ctypes.c_int
    L.v2ce_pack_weights_f16x2_wt_bytes.argtypes = [i32, i32]
    L.v2ce_pack_weights_f16x2_wt_bytes.restype = sz
    L.v2ce_conv3d_wt_variant.argtypes = [ctypes.POINTER(ConvDesc), i32, ctypes.c_char_p, sz]
    L.v2ce_conv3d_wt_variant.restype = ctypes.c_int
    L.v2ce_conv3d_head_f16x2.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 7
    L.v2ce_conv3d_head_f16x2.restype = ctypes.c_int
    L.v2ce_pack_head_weights_f16x2.argtypes = [vp, vp, vp]
    L.v2ce_pack_head_weights_f16x2.restype = ctypes.c_int
    L.v2ce_pack_head_weights_f16x2_bytes.argtypes = []
    L.v2ce_pack_head_weights_f16x2_bytes.restype = sz
    L.v2ce_absmax_batch.argtypes = [vp, i32, ctypes.c_longlong, vp, i32, vp]
    L.v2ce_absmax_batch.restype = ctypes.c_int
    L.v2ce_pack_pred_weights_f16x2.argtypes = [vp, i32, i32, vp, vp]
    L.v2ce_pack_pred_weights_f16x2.restype = ctypes.c_int
    L.v2ce_pack_pred_weights_f16x2_bytes.argtypes = []
    L.v2ce_pack_pred_weights_f16x2_bytes.restype = sz
    L.v2ce_sn_workspace_bytes.argtypes = [i32, i32]
    L.v2ce_sn_workspace_bytes.restype = sz
    L.v2ce_sn_power_iter.argtypes = [vp, vp, vp, i32, i32, vp, vp, sz, vp]
    L.v2ce_sn_batch_workspace_bytes.argtypes = [ctypes.POINTER(SnLayer), i32]
    L.v2ce_sn_batch_workspace_bytes.restype = sz
    L.v2ce_sn_update_batch.argtypes = [ctypes.POINTER(SnLayer), i32, vp, sz, vp]
    L.v2ce_sn_update_batch.restype = ctypes.c_int
    so = ctypes.POINTER(SamplerOptions)
    L.v2ce_sampler_count.argtypes = [vp, i32, i32, i32, so, vp, vp, vp]
    L.v2ce_sampler_count.restype = ctypes.c_int
    L.v2ce_sampler_pool.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    L.v2ce_sampler_pool.restype = ctypes.c_int
    L.v2ce_sampler_workspace_bytes.argtypes = [i64]
    L.v2ce_sampler_workspace_bytes.restype = sz
    L.v2ce_sampler_emit.argtypes = [vp, i32, i32, i32, so, i64, vp, vp, vp, vp, vp, sz, vp, vp]
    L.v2ce_sampler_emit.restype = ctypes.c_int
    for name in ("v2ce_ldati_count", "v2ce_ldati_emit", "v2ce_events_pack",
                 "v2ce_conv3d_fwd", "v2ce_conv3d_variant", "v2ce_pack_weights", "v2ce_sn_power_iter"):
        getattr(L, name).restype = ctypes.c_int
    _LIB = L
    return L


def source_hash() -> str:
    """16 hex digits over the kernel sources and the C ABI header of THIS tree (csrc/*.hip, csrc/*.h, include/*.h): the
    provenance stamp of the counter summaries under profiles/ -- bench.py drops a counter-derived field whose summary was
    collected from other sources (the GPU box has no .git to ask for a commit hash)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def provenance() -> dict:
    """What a profile summary records about the build it measured."""
    return {"lib_version": lib().v2ce_version().decode(), "source_hash": source_hash()}


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().v2ce_last_error().decode()
        raise V2ceHipError(f"{what} failed with code {rc}: {msg}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def require_device_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise V2ceHipError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise V2ceHipError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def conv_wt_variant(desc: ConvDesc, with_residual) -> str:
    """with_residual: False / True, or 2 for the folded-tail form."""
    buf = ctypes.create_string_buffer(96)
    check(lib().v2ce_conv3d_wt_variant(ctypes.byref(desc), int(with_residual), buf, 96), "v2ce_conv3d_wt_variant")
    return buf.value.decode()


def conv_up2_variant(desc: ConvDesc, with_shortcut: bool) -> str:
    buf = ctypes.create_string_buffer(96)
    check(lib().v2ce_conv3d_up2_variant(ctypes.byref(desc), int(with_shortcut), buf, 96), "v2ce_conv3d_up2_variant")
    return buf.value.decode()


def conv_variant(desc: ConvDesc, mapped: bool, fuse: int = 0) -> str:
    buf = ctypes.create_string_buffer(96)
    check(lib().v2ce_conv3d_variant_fused(ctypes.byref(desc), int(mapped), int(fuse), buf, 96), "v2ce_conv3d_variant")
    return buf.value.decode()
