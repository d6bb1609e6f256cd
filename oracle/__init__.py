"""CPU oracle for the V2CE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and there only as the checker.  The product package (``v2ce-toolbox_amd/``) never
imports it and fails loudly when its HIP library is missing.

Contents
--------
* ``ldati_oracle.c`` / ``ldati.py`` -- scalar C restatement of ``scripts/LDATI.py`` (stage 2).
* ``unet.py`` -- functional torch-fp32 restatement of ``scripts/v2ce_3d.py`` + ``unet_2layer.py``
  + ``submodules.py`` + ``spectral_norm.py`` (stage 1; floating point, tolerance 1e-5).
* ``glue.py`` -- restatement of the ``v2ce.py`` index arithmetic (sequence tiling, merge, offsets).
* ``make_goldens.py`` -- imports the reference from ``/root/reference`` (build container only) and
  writes the golden vectors under ``tests/golden/`` that pin all of the above.

The reference is pure Python: there is nothing to compile into ``oracle/_ref``.  Parity is pinned
by the golden vectors (outputs of the reference itself run on CPU in the build container) and by
the notebook known-answer of ``train/scripts/stage2/vis_stage2.ipynb``.
"""
