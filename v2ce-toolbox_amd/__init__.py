"""v2ce-toolbox_amd: MI355X-native V2CE hot path (V2ce3d 3-D UNet -> LDATI event sampler).

Drop-in surface (mirrors ``/root/reference``):
  * ``V2ce3d``                     -- ``scripts/v2ce_3d.py:12-30``
  * ``sample_voxel_statistical``   -- ``scripts/LDATI.py:126``
  * ``v2ce`` (CLI / glue)          -- ``v2ce.py``
All compute goes through the C-ABI HIP library ``csrc/libv2ce_hip.so`` (``include/v2ce_hip.h``);
there is no CPU fallback -- a missing library raises at first use.
"""
__all__ = ["V2ce3d", "sample_voxel_statistical", "synth", "hip"]


def __getattr__(name):
    import importlib
    if name == "V2ce3d":
        return importlib.import_module("v2ce_toolbox_amd.v2ce_3d").V2ce3d
    if name == "sample_voxel_statistical":
        return importlib.import_module("v2ce_toolbox_amd.LDATI").sample_voxel_statistical
    if name in ("synth", "hip", "LDATI", "v2ce_3d", "glue", "dist", "pipeline"):
        return importlib.import_module("v2ce_toolbox_amd." + name)
    raise AttributeError(name)
