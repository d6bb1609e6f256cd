#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE itself (imported from /root/reference) on CPU.

Runs only in the build container (the reference does not travel to the GPU box); the outputs are
small data fixtures -- inputs and the reference's outputs -- committed under tests/golden/.

    python oracle/make_goldens.py            # writes tests/golden/*.npz, *.json

Golden sets:
  G1 unet_g1.npz      V2ce3d outputs for three consecutive calls (pins spectral-norm statefulness)
                      + per-block intermediates of call #1 (forward hooks on the reference modules)
  G3 ldati_g3_*.npz   LDATI: voxels, the uniforms the reference drew (torch.rand wrapped), events
  G4 ldati_g4.json    full-size dense frame: SHA-256 of the packed event bytes + per-segment counts
  G5 ldati_kat.json   hand known-answer re-derived from the reference here
  G7 glue_g7.npz      v2ce.py video_to_voxels (center + pano) through a stub-imported v2ce.py,
                      sequence plans and per-frame offsets
  G10 sampler_g10_*   the ablation samplers of train/scripts/stage2/sample_methods (random / even baseline,
                      pure slope): voxels, every random draw of the call, events
  G8 voxelize_g8.npz  the reference voxeliser gen_discretized_event_volume (its three function
                      definitions are compiled here straight from the reference file: the module
                      itself imports h5py / numba / plotly, absent from this image) on reference
                      LDATI events

Note on G4 / sqrt: this container's torch CPU build evaluates ``torch.sqrt`` through MKL VML, which
is not correctly rounded (0.64 % of f32 inputs are 1 ulp off IEEE sqrt; measured here).  That moves
~1e-5 of the multi-event timestamps by +-1 us relative to IEEE arithmetic (which is what numpy, gcc
and the GPU's correctly-rounded sqrt compute).  G3 fixtures are chosen (asserted below) so that the
unmodified reference and the IEEE-sqrt reference agree on them exactly; G4 is produced with
``torch.sqrt`` routed through numpy's IEEE sqrt and records how many events differ from the
unmodified run.
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.environ.get("V2CE_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden")   # (V2CE_GOLDEN_DIR: the recipe test writes elsewhere)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from scripts.v2ce_3d import V2ce3d                      # noqa: E402  (reference)
import scripts.LDATI as REF_LDATI                       # noqa: E402  (reference)

from v2ce_toolbox_amd import synth                      # noqa: E402  (build-owned inputs)
from oracle import ldati as O                           # noqa: E402

torch.set_num_threads(8)


class RandCapture:
    """Wraps torch.rand to record the uniforms the reference draws (LDATI.py:171)."""

    def __enter__(self):
        self.orig = torch.rand
        self.last = None

        def wrap(*a, **k):
            r = self.orig(*a, **k)
            self.last = r.clone()
            return r
        torch.rand = wrap
        return self

    def __exit__(self, *exc):
        torch.rand = self.orig


class IeeeSqrt:
    """Routes torch.sqrt through numpy (IEEE correctly rounded) instead of MKL VML."""

    def __enter__(self):
        self.orig = torch.sqrt
        torch.sqrt = lambda x: torch.from_numpy(np.sqrt(x.numpy()))
        return self

    def __exit__(self, *exc):
        torch.sqrt = self.orig


def events_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def mt_uniforms(seed, shape):
    """CPU torch.rand(shape) after manual_seed(seed), regenerated with numpy's MT19937 (G6)."""
    mt = np.random.MT19937()
    mt._legacy_seeding(seed)
    raw = mt.random_raw(int(np.prod(shape))).astype(np.uint32)
    return ((raw & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24)).reshape(shape)


# --------------------------------------------------------------------------------------------- G1
def gen_unet():
    sd = synth.make_state_dict(0)
    model = V2ce3d()
    model.load_state_dict(sd, strict=True)
    model.eval()

    def pairs(frames):
        x = frames.astype(np.float32) / 255
        x = np.stack([x[:-1], x[1:]], axis=1)
        return ((x - np.float32(0.153)) / np.float32(0.165)).astype(np.float32)

    xa = pairs(synth.synthetic_frames(5, 16, 24, seed=11))[None]                 # [1,4,2,16,24]
    fb = synth.synthetic_frames(8, 20, 28, seed=12, pattern="noise")
    xb = np.stack([pairs(fb[0:4]), pairs(fb[4:8])])                              # [2,3,2,20,28]

    inter = {}
    hooks = []
    names = {"head": model.UNet.head}
    for i in range(4):
        names[f"enc{i}"] = model.UNet.encoders[i]
        names[f"dec{i}"] = model.UNet.decoders[i]
    for i in range(2):
        names[f"res{i}"] = model.UNet.resblocks[i]
    for n, mod in names.items():
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, n=n: inter.__setitem__(n, o.detach().clone().numpy())))
    with torch.no_grad():
        out1 = model(torch.from_numpy(xa)).contiguous().numpy()
    for h in hooks:
        h.remove()
    inter1 = dict(inter)
    with torch.no_grad():
        out2 = model(torch.from_numpy(xa)).contiguous().numpy()
        out3 = model(torch.from_numpy(xb)).contiguous().numpy()
    sd_after = model.state_dict()
    extra = {"u_after3_" + k: v.numpy() for k, v in sd_after.items()
             if k.endswith("weight_u") and ("resblocks.0.conv1" in k or "decoders.3.conv2" in k)}
    np.savez_compressed(os.path.join(GOLD, "unet_g1.npz"), xa=xa, xb=xb, out1=out1, out2=out2,
                        out3=out3, **{"inter_" + k: v for k, v in inter1.items()}, **extra)
    print("G1: out1 max %.4f  frac>0 %.3f  frac>1 %.4f | |out2-out1| max %.3e" %
          (out1.max(), (out1 > 0).mean(), (out1 > 1).mean(), np.abs(out2 - out1).max()))


# --------------------------------------------------------------------------------------------- G3
def run_ref_ldati(vox, fps, t0, seed, ieee_sqrt=False, strategy="slope", **opts):
    with RandCapture() as cap:
        torch.manual_seed(seed)
        kw = dict(t0=t0, fps=fps, additional_events_strategy=strategy, **opts)
        if ieee_sqrt:
            with IeeeSqrt():
                res = REF_LDATI.sample_voxel_statistical(torch.from_numpy(vox), **kw)
        else:
            res = REF_LDATI.sample_voxel_statistical(torch.from_numpy(vox), **kw)
        u = cap.last.numpy()
    return res, u


def gen_ldati_small():
    cases = {
        "sparse": (synth.synthetic_voxels(3, 12, 14, seed=21, regime="sparse"), 30, 0),
        "frac": (synth.synthetic_voxels(3, 12, 14, seed=22, regime="frac"), 25, 0),
        "stress": (synth.synthetic_voxels(2, 12, 14, seed=23, regime="stress"), 30, 0),
        "t0fps60": (synth.synthetic_voxels(2, 9, 11, seed=24, regime="stress"), 60, 0.5),
        "ragged": (synth.synthetic_voxels(1, 1, 67, seed=25, regime="stress"), 24, 0),
        # a time bin of 11 111 us: more keys than the sweep kernel's LDS histogram holds (two-level path only)
        "fps10": (synth.synthetic_voxels(2, 9, 11, seed=27, regime="stress"), 10, 0),
    }
    # additional_events_strategy='none' (only single-event voxels emit, LDATI.py:206-207,241)
    cases["none"] = (synth.synthetic_voxels(2, 12, 14, seed=26, regime="stress"), 30, 0)
    for name, (vox, fps, t0) in cases.items():
        seed = 100 + len(name)
        strategy = "none" if name == "none" else "slope"
        res, u = run_ref_ldati(vox, fps, t0, seed, strategy=strategy)
        res_ieee, _ = run_ref_ldati(vox, fps, t0, seed, ieee_sqrt=True, strategy=strategy)
        assert all(events_equal(a, b) for a, b in zip(res, res_ieee)), \
            f"{name}: MKL-VML sqrt and IEEE sqrt disagree on this fixture; pick another seed"
        B, _, _, H, W = vox.shape
        u = u.reshape(B, 2, 9, H, W, -1)
        lens = np.array([len(r) for r in res], np.int64)
        ev = np.concatenate([np.asarray(r) for r in res]) if lens.sum() else \
            np.empty(0, O.EVENT_DTYPE)
        assert ev.dtype.itemsize == 13
        np.savez_compressed(os.path.join(GOLD, f"ldati_g3_{name}.npz"), vox=vox, uniforms=u,
                            fps=np.float64(fps), t0=np.float64(t0), lens=lens,
                            events=np.frombuffer(ev.tobytes(), np.uint8))
        print(f"G3 {name}: B={B} HxW={H}x{W} fps={fps} t0={t0} max_n={u.shape[-1]} events={lens}")


def gen_ldati_options():
    """G3 for the remaining option values of sample_voxel_statistical (LDATI.py:126):
    bidirectional relocation, pooled slope, 'random' strategy."""
    cases = {
        "bidir": (synth.synthetic_voxels(2, 12, 14, seed=31, regime="stress"), 30, 0, dict(bidirectional=True)),
        "bidir_sparse": (synth.synthetic_voxels(3, 12, 14, seed=32, regime="sparse"), 25, 0.25, dict(bidirectional=True)),
        "avg3": (synth.synthetic_voxels(2, 12, 14, seed=33, regime="stress"), 30, 0, dict(pooling_type="avg")),
        "avg5": (synth.synthetic_voxels(2, 9, 11, seed=34, regime="stress"), 60, 0,
                 dict(pooling_type="avg", pooling_kernel_size=5)),
        "weighted": (synth.synthetic_voxels(2, 12, 14, seed=35, regime="stress"), 30, 0, dict(pooling_type="weighted")),
        "random": (synth.synthetic_voxels(2, 12, 14, seed=36, regime="stress"), 30, 0,
                   dict(additional_events_strategy="random")),
        "bidir_weighted": (synth.synthetic_voxels(2, 12, 14, seed=37, regime="stress"), 30, 0,
                           dict(bidirectional=True, pooling_type="weighted")),
    }
    for name, (vox, fps, t0, opts) in cases.items():
        opts = dict(opts)
        strategy = opts.pop("additional_events_strategy", "slope")
        for seed in range(200, 260):
            res, u = run_ref_ldati(vox, fps, t0, seed, strategy=strategy, **opts)
            res_ieee, _ = run_ref_ldati(vox, fps, t0, seed, ieee_sqrt=True, strategy=strategy, **opts)
            if all(events_equal(a, b) for a, b in zip(res, res_ieee)):
                break
        else:
            raise AssertionError(f"{name}: MKL-VML sqrt and IEEE sqrt disagree for every seed tried")
        B, _, _, H, W = vox.shape
        u = u.reshape(B, 2, 9, H, W, -1)
        lens = np.array([len(r) for r in res], np.int64)
        ev = np.concatenate([np.asarray(r) for r in res])
        np.savez_compressed(os.path.join(GOLD, f"ldati_g3_opt_{name}.npz"), vox=vox, uniforms=u,
                            fps=np.float64(fps), t0=np.float64(t0), lens=lens,
                            events=np.frombuffer(ev.tobytes(), np.uint8),
                            strategy=np.array(strategy), bidirectional=np.array(bool(opts.get("bidirectional", False))),
                            pooling_type=np.array(opts.get("pooling_type", "none")),
                            pooling_kernel_size=np.array(int(opts.get("pooling_kernel_size", 3))))
        print(f"G3 option {name}: seed {seed} B={B} HxW={H}x{W} fps={fps} t0={t0} max_n={u.shape[-1]} events={lens}")


# --------------------------------------------------------------------------------------------- G10
class DrawCapture:
    """Records every torch.rand of the reference call and routes torch.bernoulli(p) through a recorded
    uniform draw (u < p: what torch's CPU kernel computes per element from its own generator), so that
    all random draws of a sampler call become data of the fixture."""

    def __enter__(self):
        self.rand0, self.bern0 = torch.rand, torch.bernoulli
        self.rands, self.berns = [], []

        def rand(*a, **k):
            r = self.rand0(*a, **k)
            self.rands.append(r.clone())
            return r

        def bern(p):
            u = self.rand0(p.shape)
            self.berns.append(u.clone())
            return (u < p).to(p.dtype)
        torch.rand, torch.bernoulli = rand, bern
        return self

    def __exit__(self, *exc):
        torch.rand, torch.bernoulli = self.rand0, self.bern0


def reference_sample_methods():
    """The two ablation samplers, imported from the reference (h5py, which random_even_sample.py imports
    and never uses, is absent here: stubbed)."""
    sys.path.insert(0, "/root/reference/train/scripts/stage2")
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    from sample_methods import random_even_sample as RE, pure_slope_sample as PS
    return RE, PS


def gen_sample_methods():
    """G10: sample_voxel_baseline (random / even) and the pure-slope sampler."""
    from oracle import sample_methods as OS
    RE, PS = reference_sample_methods()
    cases = {
        "random": ("baseline", synth.synthetic_voxels(2, 12, 14, seed=41, regime="stress"), 30, 0, dict(random=True)),
        "random_sparse": ("baseline", synth.synthetic_voxels(3, 9, 11, seed=42, regime="sparse"), 25, 0.25, dict(random=True)),
        "even": ("baseline", synth.synthetic_voxels(2, 12, 14, seed=43, regime="stress"), 30, 0, dict(even=True)),
        "even_frac": ("baseline", synth.synthetic_voxels(2, 9, 11, seed=44, regime="frac"), 60, 0, dict(even=True)),
        "slope": ("pure_slope", synth.synthetic_voxels(2, 12, 14, seed=45, regime="stress"), 30, 0, {}),
        "slope_sparse": ("pure_slope", synth.synthetic_voxels(3, 9, 11, seed=46, regime="sparse"), 25, 0.25, {}),
        "slope_ragged": ("pure_slope", synth.synthetic_voxels(1, 1, 67, seed=47, regime="stress"), 24, 0, {}),
    }
    for name, (kind, vox, fps, t0, opts) in cases.items():
        B, _, _, H, W = vox.shape

        def run(seed, ieee):
            with DrawCapture() as cap:
                torch.manual_seed(seed)
                y = torch.from_numpy(vox.copy())
                fn = (lambda: RE.sample_voxel_baseline(y, t0=t0, fps=fps, **opts)) if kind == "baseline" else \
                     (lambda: PS.sample_voxel_statistical(y, t0=t0, fps=fps))
                if ieee:
                    with IeeeSqrt():
                        res = fn()
                else:
                    res = fn()
            return res, cap

        for seed in range(300, 360):
            res, cap = run(seed, False)
            res_ieee, _ = run(seed, True)
            if all(events_equal(a, b) for a, b in zip(res, res_ieee)):
                break
        else:
            raise AssertionError(f"{name}: MKL-VML sqrt and IEEE sqrt disagree for every seed tried")
        # the Bernoulli draws arrive plane by plane in pick_and_sort's order: frame, bin, negative (P index 1) first
        u_bern = np.empty((B, 2, 10, H, W), np.float32)
        it = iter(cap.berns)
        for b in range(B):
            for c in range(10):
                for pi in (1, 0):
                    u_bern[b, pi, c] = next(it).numpy()
        rands = [r.numpy() for r in cap.rands]
        if kind == "baseline" and opts.get("even"):
            assert not rands
            u_int, u_dec = np.zeros((B, 2, 10, H, W, 0), np.float32), np.zeros((B, 2, 10, H, W), np.float32)
        elif kind == "baseline":
            u_int, u_dec = rands[0].reshape(B, 2, 10, H, W, -1), rands[1].reshape(B, 2, 10, H, W)
        else:
            u_dec, u_int = rands[0].reshape(B, 2, 10, H, W), rands[1].reshape(B, 2, 10, H, W, -1)
        lens = np.array([len(r) for r in res], np.int64)
        ev = np.concatenate([np.asarray(r) for r in res])
        assert ev.dtype.itemsize == 13
        # the oracle is pinned right here as well
        if kind == "baseline":
            mine = OS.sample_voxel_baseline(vox, t0, fps, u_int=u_int, u_dec=u_dec, u_bern=u_bern, **opts)
        else:
            mine = OS.sample_voxel_pure_slope(vox, t0, fps, u_int=u_int, u_dec=u_dec, u_bern=u_bern)
        assert all(events_equal(a, b) for a, b in zip(res, mine)), name
        np.savez_compressed(os.path.join(GOLD, f"sampler_g10_{name}.npz"), vox=vox, kind=np.array(kind),
                            mode=np.array("even" if opts.get("even") else "random" if opts.get("random") else "slope"),
                            fps=np.float64(fps), t0=np.float64(t0), u_int=u_int, u_dec=u_dec, u_bern=u_bern,
                            lens=lens, events=np.frombuffer(ev.tobytes(), np.uint8))
        print(f"G10 {name}: seed {seed} B={B} HxW={H}x{W} fps={fps} t0={t0} M={u_int.shape[-1]} events={lens}")



def gen_sample_methods_pooled():
    """G10 (pooled): the pure-slope sampler with pooling_type 'weighted' / 'avg' (pure_slope_sample.py:79-85).  The pooled
    values are sums of non-integer f32 values (torch CPU conv2d / avg_pool2d): the oracle restates them with a fixed
    row-major summation order and is required to be CLOSE (same events, timestamps within 1 us), not bit-equal."""
    from oracle import sample_methods as OS
    _, PS = reference_sample_methods()
    cases = {
        "slope_weighted": (synth.synthetic_voxels(2, 12, 14, seed=48, regime="stress"), 30, 0, dict(pooling_type="weighted")),
        "slope_avg3": (synth.synthetic_voxels(2, 9, 11, seed=49, regime="sparse"), 25, 0.25, dict(pooling_type="avg", pooling_kernel_size=3)),
        "slope_avg5": (synth.synthetic_voxels(1, 12, 14, seed=50, regime="stress"), 30, 0, dict(pooling_type="avg", pooling_kernel_size=5)),
    }
    for name, (vox, fps, t0, opts) in cases.items():
        B, _, _, H, W = vox.shape
        seed = 400
        with DrawCapture() as cap:
            torch.manual_seed(seed)
            y = torch.from_numpy(vox.copy())
            with IeeeSqrt():
                res = PS.sample_voxel_statistical(y, t0=t0, fps=fps, **opts)
        u_bern = np.empty((B, 2, 10, H, W), np.float32)
        it = iter(cap.berns)
        for b in range(B):
            for c in range(10):
                for pi in (1, 0):
                    u_bern[b, pi, c] = next(it).numpy()
        rands = [r.numpy() for r in cap.rands]
        u_dec, u_int = rands[0].reshape(B, 2, 10, H, W), rands[1].reshape(B, 2, 10, H, W, -1)
        lens = np.array([len(r) for r in res], np.int64)
        ev = np.concatenate([np.asarray(r) for r in res])
        mine = OS.sample_voxel_pure_slope(vox, t0, fps, u_int=u_int, u_dec=u_dec, u_bern=u_bern, **opts)
        diffs = [OS.events_close(np.asarray(a), np.asarray(b)) for a, b in zip(res, mine)]
        assert all(d >= 0 for d in diffs), (name, diffs)
        np.savez_compressed(os.path.join(GOLD, f"sampler_g10p_{name}.npz"), vox=vox, fps=np.float64(fps), t0=np.float64(t0),
                            pooling_type=np.array(opts["pooling_type"]), pooling_kernel_size=np.array(int(opts.get("pooling_kernel_size", 3))),
                            u_int=u_int, u_dec=u_dec, u_bern=u_bern, lens=lens, events=np.frombuffer(ev.tobytes(), np.uint8))
        print(f"G10 pooled {name}: B={B} HxW={H}x{W} fps={fps} t0={t0} M={u_int.shape[-1]} events={lens} "
              f"timestamps differing from the oracle by 1 us: {diffs}")


# --------------------------------------------------------------------------------------------- G4
def gen_ldati_large():
    H, W, seed = 260, 346, 4242
    vox = synth.synthetic_voxels(1, H, W, seed=44, regime="stress")
    res_raw, u = run_ref_ldati(vox, 30, 0, seed)
    res, u2 = run_ref_ldati(vox, 30, 0, seed, ieee_sqrt=True)
    assert np.array_equal(u, u2)
    max_n = u.reshape(1, 2, 9, H, W, -1).shape[-1]
    assert np.array_equal(mt_uniforms(seed, (1, 2, 9, H, W, max_n)).reshape(u.shape), u), "G6"
    ev, raw = np.asarray(res[0]), np.asarray(res_raw[0])
    seg, mx = O.count(vox)
    assert mx == max_n and int(seg.sum()) == len(ev)
    assert int(seg.min()) >= 32768, "every segment must take the stable-sort path"
    key = lambda a: a[np.lexsort((a["timestamp"], a["x"], a["y"], a["polarity"]))]
    ndiff = int((key(ev) != key(raw)).sum())
    meta = {
        "H": H, "W": W, "fps": 30, "t0": 0, "vox_seed": 44, "vox_regime": "stress",
        "torch_seed": seed, "max_n": int(max_n), "num_events": int(len(ev)),
        "seg_counts": seg.reshape(-1).tolist(),
        "sha256_packed_events": hashlib.sha256(ev.tobytes()).hexdigest(),
        "sha256_timestamps": hashlib.sha256(np.ascontiguousarray(ev["timestamp"]).tobytes()).hexdigest(),
        "events_differing_from_mkl_vml_sqrt_run": ndiff,
        "first_events": [[int(v) for v in e] for e in ev[:8].tolist()],
        "last_events": [[int(v) for v in e] for e in ev[-4:].tolist()],
    }
    with open(os.path.join(GOLD, "ldati_g4.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("G4:", {k: meta[k] for k in ("num_events", "max_n", "events_differing_from_mkl_vml_sqrt_run")})


# --------------------------------------------------------------------------------------------- G5
def gen_kat():
    y = np.zeros((1, 2, 10, 1, 2), np.float32)
    y[0, 0, :, 0, 0] = [.3, .4, .5, 0, 0, 1.2, 0, 0, .9, .6]
    y[0, 0, :, 0, 1] = [2, 3.5, 0, 0, 4, 1, 0, 0, 0, 2.7]
    y[0, 1, :, 0, 0] = [0, 0, 1, 1, 1, 0, 0, 2.5, .25, .25]
    res, u = run_ref_ldati(y, 30, 0, 0)
    ev = np.asarray(res[0])
    n, _ = REF_LDATI.y_relocate(torch.from_numpy(y).reshape(2, 10, 1, 2))
    # notebook known-answer (train/scripts/stage2/vis_stage2.ipynb cells 1-2): the voxel printed in
    # cell 1 and the three deterministic single-event times printed in cell 2 (units of 1/300 s)
    nb_vox = np.zeros((1, 2, 10, 1, 1), np.float32)
    nb_vox[0, 0, :, 0, 0] = [0, 0, 0.9179, 0.0821, 0.9962, 0.0038, 0.5287, 2.8454, 1.6884, 0.9375]
    nb_res, _ = run_ref_ldati(nb_vox, 30, 0, 0)
    nb_ts = np.sort(np.asarray(nb_res[0])["timestamp"]) * (300 / 1e6)
    # empty input (all-zero voxels => max_n == 0): the reference raises from its debug logging
    # (LDATI.py:200 torch.max of an empty tensor)
    try:
        run_ref_ldati(np.zeros((2, 2, 10, 5, 6), np.float32), 30, 0, 0)
        empty_behaviour = "returns"
    except RuntimeError as e:
        empty_behaviour = "RuntimeError: " + str(e).split(".")[0]
    meta = {
        "empty_input": empty_behaviour,
        "hand": {"vox": y.reshape(-1).tolist(), "shape": list(y.shape),
                 "uniforms": u.reshape(-1).tolist(), "uniforms_shape": [1, 2, 9, 1, 2, int(u.size // 36)],
                 "counts": n.numpy().reshape(-1).tolist(),
                 "events": [[int(v) for v in e] for e in ev.tolist()]},
        "notebook": {"vox": nb_vox.reshape(-1).tolist(),
                     "printed_first_three": [2.3133, 4.4484, 7.1901],
                     "reference_first_three_here": nb_ts[:3].tolist(), "num_events": int(len(nb_ts))},
    }
    with open(os.path.join(GOLD, "ldati_kat.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("G5: hand KAT events", len(ev), "| notebook first three", nb_ts[:3], "n =", len(nb_ts))


# --------------------------------------------------------------------------------------------- G7
_REF_V2CE = None


def import_reference_v2ce():
    """Import /root/reference/v2ce.py with its missing third-party imports stubbed.  Imported ONCE: a second
    call used to install a fresh cv2 stub in sys.modules while the cached reference module kept the first one, so
    gen_event_frames() patched a stub the reference never saw (the default invocation died there)."""
    global _REF_V2CE
    if _REF_V2CE is not None:
        return _REF_V2CE
    cv2 = types.ModuleType("cv2")

    def resize(img, size):
        assert (img.shape[1], img.shape[0]) == tuple(size), "stub resize: identity only"
        return img
    cv2.resize = resize
    cv2.IMREAD_GRAYSCALE = 0
    sys.modules["cv2"] = cv2
    pl = types.ModuleType("pathlib2")
    import pathlib
    pl.Path = pathlib.Path
    sys.modules["pathlib2"] = pl
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean), torch.tensor(std)

        def __call__(self, t):     # torchvision F.normalize: tensor.sub_(mean).div_(std), f32
            return t.clone().sub_(self.mean.view(-1, 1, 1)).div_(self.std.view(-1, 1, 1))

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    tr.Normalize, tr.Compose = Normalize, Compose
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr
    import v2ce as ref_v2ce
    ref_v2ce.logger = logging.getLogger("V2CE")
    torch.Tensor.cuda = lambda self, *a, **k: self
    _REF_V2CE = ref_v2ce
    return ref_v2ce


def gen_glue():
    ref = import_reference_v2ce()
    H, WF, width, N, bs = 8, 20, 12, 20, 2
    frames = synth.synthetic_frames(N, H, WF, seed=31)

    class FakeCap:
        frame_count = N

        def read_frames_at_indices(self, idx):
            return frames[list(idx)]

    outs = {}
    for infer_type in ("center", "pano"):
        model = V2ce3d()
        model.load_state_dict(synth.make_state_dict(0), strict=True)
        model.eval()
        outs[infer_type] = ref.video_to_voxels(model, vidcap=FakeCap(), infer_type=infer_type,
                                               seq_len=16, width=width, height=H, batch_size=bs)
    pre = ref.image_pre_processing(frames[:5], height=H).numpy()
    plans = {}
    for n in (17, 18, 33, 100, 2048):
        sequence_num = np.ceil((n - 1) / 16).astype(int)
        mode = (n - 1) % 16
        starts = np.arange(sequence_num) * 16
        if mode != 0:
            starts[-1] -= (16 - mode)
        plans[f"plan_{n}"] = np.concatenate([[sequence_num, mode], starts]).astype(np.int64)
    offs = {f"offsets_fps{fps}": np.array([int(i * 1 / fps * 1e6) for i in range(4096)], np.int64)
            for fps in (25, 30)}
    np.savez_compressed(os.path.join(GOLD, "glue_g7.npz"), frames=frames, pre5=pre,
                        center=outs["center"], pano=outs["pano"],
                        params=np.array([H, WF, width, N, bs], np.int64), **plans, **offs)
    print("G7: center", outs["center"].shape, "pano", outs["pano"].shape)


# --------------------------------------------------------------------------------------------- G9
def gen_event_frames():
    """The frames the reference's write_event_frame_video (v2ce.py:241-280) hands to cv2.VideoWriter, with
    the container replaced by a recorder (cv2 is absent here; cvtColor RGB2BGR = channel reversal)."""
    ref = import_reference_v2ce()
    cv2 = sys.modules["cv2"]
    got = []

    class Recorder:
        def __init__(self, path, fourcc, fps, size):
            got.append({"fps": fps, "size": size, "frames": []})

        def write(self, frame):
            got[-1]["frames"].append(frame.copy())

        def release(self):
            pass
    cv2.VideoWriter = Recorder
    cv2.VideoWriter_fourcc = lambda *a: 0
    cv2.COLOR_RGB2BGR = 4
    cv2.cvtColor = lambda img, code: img[..., ::-1]
    rng = np.random.default_rng(9)
    vox = (rng.gamma(0.3, 1.2, (6, 2, 10, 9, 14)) * (rng.random((6, 2, 10, 9, 14)) < 0.4)).astype(np.float32)
    out = {"vox": vox}
    for name, keep, ceil, pct in (("rgb", True, 10, 98), ("gray", False, 10, 98), ("rgb_ceil", True, 2, 90)):
        ref.write_event_frame_video(vox, "unused.mp4", 30, ceil, pct, keep)
        out[f"bgr_{name}"] = np.stack(got[-1]["frames"])
        out[f"args_{name}"] = np.array([int(keep), ceil, pct], np.int64)
        assert got[-1]["size"] == (14, 9)
    np.savez_compressed(os.path.join(GOLD, "event_frames_g9.npz"), **out)
    print("G9:", {k: v.shape for k, v in out.items()})


# --------------------------------------------------------------------------------------------- G8
def reference_voxelizer():
    """gen_discretized_event_volume + its two helpers, compiled from the reference source file at
    generation time (nothing of it is stored in this repo)."""
    import ast
    path = os.path.join("/root/reference", "train", "scripts", "utils", "events_utils.py")
    tree = ast.parse(open(path).read())
    want = {"calc_floor_ceil_delta", "create_update", "gen_discretized_event_volume"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want],
                     type_ignores=[])
    assert {n.name for n in mod.body} == want
    ns = {"torch": torch, "np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns["gen_discretized_event_volume"]


def gen_voxelize():
    from oracle.voxelize import gen_discretized_event_volume as oracle_vox
    ref = reference_voxelizer()
    out = {}
    for name in ("stress", "sparse"):
        z = np.load(os.path.join(GOLD, f"ldati_g3_{name}.npz"), allow_pickle=True)
        H, W = z["vox"].shape[-2:]
        lens = z["lens"]
        ev = z["events"].view(O.EVENT_DTYPE)[:int(lens[0])]  # the events of frame-pair 0 (the reference's unit)
        vol = ref(ev.copy(), (20, H, W)).numpy()
        mine = oracle_vox(ev, (20, H, W))
        assert np.array_equal(vol, mine), f"oracle voxeliser differs from the reference on {name}"
        out[f"events_{name}"] = ev
        out[f"volume_{name}"] = vol
        print(f"G8 {name}: {len(ev)} events, sum {vol.sum():.3f}")
    np.savez_compressed(os.path.join(GOLD, "voxelize_g8.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["unet", "ldati", "options", "large", "kat", "glue", "voxelize", "event_frames", "samplers"]
    # dependency order: the voxeliser golden (G8) is made from the events of the LDATI goldens (G3), so "ldati" comes first --
    # the default order works from an EMPTY tests/golden/ (tests/test_oracle_goldens_recipe.py runs it into a temp directory)
    if "unet" in which:
        gen_unet()
    if "ldati" in which:
        gen_ldati_small()
    if "options" in which:
        gen_ldati_options()
    if "large" in which:
        gen_ldati_large()
    if "kat" in which:
        gen_kat()
    if "glue" in which:
        gen_glue()
    if "voxelize" in which:
        gen_voxelize()
    if "event_frames" in which:
        gen_event_frames()
    if "samplers" in which:
        gen_sample_methods()
        gen_sample_methods_pooled()
    if "samplers_pooled" in which:
        gen_sample_methods_pooled()
