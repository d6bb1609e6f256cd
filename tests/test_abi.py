"""CPU: the C-ABI library loads and exports every symbol include/v2ce_hip.h declares; argument
validation that needs no GPU; the product has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

from v2ce_toolbox_amd import hip


def declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "v2ce_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(v2ce_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = hip.lib()
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), s
    assert set(syms) == set(hip.EXPORTS)
    assert b"gfx950" in L.v2ce_version()


def test_argument_validation_without_gpu():
    L = hip.lib()
    assert L.v2ce_ldati_count(None, 1, 4, 4, None, None, 0, None, None, None) == -1
    assert b"null" in L.v2ce_last_error()
    assert L.v2ce_ldati_lds_bytes(30.0, 0.0) > 0
    assert L.v2ce_ldati_lds_bytes(5.0, 0.0) == 0          # bin too wide for the sweep kernel's LDS histogram ...
    assert L.v2ce_ldati_workspace_bytes(1, 260, 346, 5.0, 0.0, None, 100000, 20000, 500, 1) > 0     # ... the two-level path runs
    assert L.v2ce_ldati_workspace_bytes(1, 260, 346, 30.0, 0.0, None, 10 ** 7, 10 ** 6, 20000, 1) == 0   # tile-bin beyond LDS
    rnd = hip.LdatiOptions(strategy=hip.STRATEGY_RANDOM, bidirectional=0, pooling_type=0, pooling_kernel_size=3)
    assert L.v2ce_ldati_workspace_bytes(1, 260, 346, 30.0, 0.0, ctypes.byref(rnd), 100000, 20000, 500, 1) > 16 * 100000
    # tile counts [B][T][9] + tile offsets [B][T][9] + the offsets again as one row of T (padded to 8) per segment (round 4)
    assert L.v2ce_ldati_tile_ws_bytes(24, 260, 346) == 2 * 24 * 88 * 9 * 4 + 24 * 9 * 88 * 4
    assert L.v2ce_sn_workspace_bytes(512, 13824) >= 4 * (512 + 13824)
    d = hip.ConvDesc(B=1, T=16, C0=64, H0=130, W0=173, C1=0, Hin=130, Win=173, Cout=64, Hout=130,
                     Wout=173, ksize=5, stride_hw=1, act=1, tile_t=0, tile_h=0, tile_w=0, precision=0)
    buf = ctypes.create_string_buffer(64)
    assert L.v2ce_conv3d_variant(ctypes.byref(d), 0, buf, 64) == -2     # ksize 5 unsupported
    d.ksize = 3
    assert L.v2ce_conv3d_variant(ctypes.byref(d), 0, buf, 64) == 0
    assert buf.value.startswith(b"conv3d_kernel<3,1,")


def test_no_cpu_fallback():
    from v2ce_toolbox_amd.LDATI import sample_voxel_statistical
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    with pytest.raises(hip.V2ceHipError):
        sample_voxel_statistical(torch.zeros(1, 2, 10, 4, 4))
    m = V2ce3d().eval()
    with pytest.raises(hip.V2ceHipError):
        m(torch.zeros(1, 2, 2, 8, 8))
    with pytest.raises(NotImplementedError):
        m.train()


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "v2ce-toolbox_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libv2ce_oracle" not in src, f
