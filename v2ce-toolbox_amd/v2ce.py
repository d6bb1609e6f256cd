"""Command line + driver of the hot path: drop-in for ``/root/reference/v2ce.py``.

Flags, defaults, output file name (``{name}-ceil_{ceil}-fps_{fps}[-suffix]-events.npz``), npz key
(``event_stream``) and record dtype are the reference's (v2ce.py:282-302,317,371-372).  Added flags:
``--npy_frames`` / ``--synthetic`` (frame sources that need no OpenCV), ``--device``, ``--seed``,
``--rng``.  Unlike the reference the voxel grid never leaves the device between the model and
LDATI, the whole clip goes through LDATI in chunks on the device, and the per-frame timestamp
offset (v2ce.py:365) is fused into the emit kernel.

Under ``torchrun`` (WORLD_SIZE > 1) the sequences of every batch are shared out over the ranks (``-b 32`` on
8 GPUs: four sequences per GPU per model call; pano with a world size that is a multiple of the tile count:
one tile per GPU + all-to-all re-shard, ``pipeline.py``) and the packed events stream to rank 0 over RCCL
batch by batch (``dist.py``); rank 0 writes the file.
"""
from __future__ import annotations

import argparse
import logging
import os
import os.path as op
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from . import dist as vdist
from . import glue
from .LDATI import EVENT_DTYPE, ldati_device
from .v2ce_3d import V2ce3d

logger = logging.getLogger("V2CE")


def SBool(v):
    """v2ce.py:19-27."""
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    elif v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_trained_mode(model_path="./weights/v2ce_3d.pt", device="cuda", precision="f16x2"):
    """v2ce.py:30-43."""
    model = V2ce3d(precision=precision)
    model.load_state_dict(torch.load(model_path, map_location="cpu"))
    model = model.eval()
    return model.to(device)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--fps", type=int, default=30, help="FPS of the output video")
    p.add_argument("--seq_len", type=int, default=16, help="Sequence length")
    p.add_argument("--ceil", type=int, default=10, help="The ceiling of the ef value")
    p.add_argument("-u", "--upper_bound_percentile", type=int, default=98)
    p.add_argument("-f", "--image_folder", type=str, help="The folder containing the images to infer")
    p.add_argument("-i", "--input_video_path", type=str, help="The path to the input video")
    p.add_argument("-o", "--out_folder", type=str, default="./output")
    p.add_argument("-t", "--infer_type", type=str, default="center", help="center or pano")
    p.add_argument("-m", "--model_path", type=str, default="./weights/v2ce_3d.pt")
    p.add_argument("--out_name_suffix", type=str, default="")
    p.add_argument("--max_frame_num", type=int, default=1800)
    p.add_argument("--width", type=int, default=346)
    p.add_argument("--height", type=int, default=260)
    p.add_argument("--write_event_frame_video", type=SBool, default=True, nargs="?", const=True)
    p.add_argument("--vis_keep_polarity", type=SBool, default=True, nargs="?", const=True)
    p.add_argument("-l", "--log_level", type=str, default="info")
    p.add_argument("-b", "--batch_size", type=int, default=1, help="Batch size for inference")
    p.add_argument("--stage2_batch_size", type=int, default=24)
    # additions of this build
    p.add_argument("--npy_frames", type=str, help="uint8 [N,H,W] grayscale frames in a .npy file")
    p.add_argument("--synthetic", type=int, default=0, help="generate N synthetic frames instead of reading")
    p.add_argument("--synthetic_weights", type=int, default=None,
                   help="seed of synthetic weights (the pretrained file is a separate download)")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--seed", type=int, default=0, help="Philox seed of the LDATI draws")
    p.add_argument("--rng", type=str, default="philox", choices=["philox", "torch"])
    p.add_argument("--stream_events", type=SBool, default=True, nargs="?", const=True,
                   help="write the events file while the clip runs (batch by batch, same np.load content) instead of one "
                        "np.savez of the whole clip at the end")
    p.add_argument("--precision", type=str, default="f16x2", choices=["f16x2", "f32"],
                   help="stage-1 conv arithmetic: split-half fp16 MFMA (f32-equivalent accuracy) or exact f32 MFMA")
    return p


def read_image_folder(folder, max_frame_num):
    """v2ce.py:325-326,172-174: sorted *.png, read as grayscale: with ``cv2.imread(p, IMREAD_GRAYSCALE)`` like the
    reference where OpenCV is installed, else with PIL.  8-bit grayscale files give identical arrays either way;
    COLOUR files differ by up to +-1 grey level (OpenCV's PNG reader converts with libpng's rgb_to_gray, PIL with
    the ITU-R 601 integer transform), which the model turns into ~1e-3 voxel differences: convert such folders to
    grayscale once if bit-reproducibility against a cv2 machine matters."""
    paths = sorted(op.join(folder, f) for f in os.listdir(folder) if f.endswith(".png"))[:max_frame_num]
    logger.info(f"Now processing {folder}, Found {len(paths)} images.")
    try:
        import cv2
        return np.stack([cv2.imread(p, cv2.IMREAD_GRAYSCALE) for p in paths], axis=0)
    except ImportError:
        from PIL import Image
        return np.stack([np.asarray(Image.open(p).convert("L")) for p in paths], axis=0)


def read_video(path, max_frame_num):
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("reading a video container needs OpenCV (scripts/video_reader.py of the "
                           "reference); it is not installed here -- use -f, --npy_frames or "
                           "--synthetic") from e
    cap = cv2.VideoCapture(path)
    frames = []
    while len(frames) < max_frame_num:
        ok, fr = cap.read()
        if not ok:
            break
        frames.append(cv2.cvtColor(fr, cv2.COLOR_BGR2GRAY))
    return np.stack(frames, axis=0)


def events_from_voxels(pred_voxel: torch.Tensor, fps, stage2_batch_size=24, seed=0, rng="philox",
                       first_pair=0):
    """v2ce.py:351-367 on the device: LDATI in chunks of frame-pairs with the per-frame offset
    int(i*1/fps*1e6) (global pair index) fused.  Returns the list of packed uint8 record buffers
    (device) and the per-frame event counts.

    With the counter-based Philox draws the result does not depend on the chunking, so chunks are
    at least 96 pairs (fewer host synchronisations); with rng='torch' the chunk size shapes the dense
    uniform tensor exactly like the reference's --stage2_batch_size (LDATI.py:169-171)."""
    L = pred_voxel.shape[0]
    chunk = stage2_batch_size if rng == "torch" else max(stage2_batch_size, 96)
    packed, counts, status = [], [], None
    for i in range(0, L, chunk):
        part = pred_voxel[i:i + chunk]
        add = torch.tensor([glue.frame_offset_us(first_pair + i + j, fps) for j in range(part.shape[0])],
                           dtype=torch.int64, device=part.device)
        ev = ldati_device(part, fps=fps, rng=rng, seed=seed, frame_base=first_pair + i, frame_ts_add=add)
        packed.append(ev.packed())
        counts.append(ev.frame_counts)
        if ev._status is not None:                             # device status words of the chunks, folded on the stream
            status = ev._status.clone() if status is None else torch.maximum(status, ev._status)
    if status is not None and int(status.item()) != 0:         # (the caller downloads next: this wait costs nothing extra)
        from . import hip
        raise hip.V2ceHipError("LDATI: a (frame, bin) segment could not be ordered on the device (more equal-time events "
                               "than the LDS sort holds at an fps beyond the sweep kernel's histogram)")
    return packed, np.concatenate(counts)


def download_events(packed_list) -> np.ndarray:
    """One D2H pass of the packed records into a pinned host buffer, viewed as the structured
    array of LDATI.py:308 (no host-side concatenation)."""
    total = sum(int(p.numel()) for p in packed_list)
    if not packed_list or not packed_list[0].is_cuda:          # CPU stand-ins (tests)
        return np.ascontiguousarray(torch.cat(list(packed_list)).numpy()).view(EVENT_DTYPE)
    host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    off = 0
    for p in packed_list:
        n = int(p.numel())
        host[off:off + n].copy_(p, non_blocking=True)
        off += n
    torch.cuda.synchronize(packed_list[0].device)
    return host.numpy().view(EVENT_DTYPE)


def event_frame_images(efs: np.ndarray, ceil, upper_bound_percentile=98, keep_polarity=True) -> np.ndarray:
    """v2ce.py:253-269,275-276: efs [L,3,H,W] = pipeline.event_frame_sums (per-polarity sums over the ten
    bins, and the sum of all twenty planes) -> uint8 RGB frames [L,H,W,3]: the two polarities in red and
    green (or their sum in grey), clipped at min(percentile of the non-zero values, ceil), scaled to 255."""
    L, _, H, W = efs.shape
    if keep_polarity:
        efs = np.concatenate([efs[:, :2], np.zeros((L, 1, H, W))], axis=1)   # float64 like the reference's zeros
    else:
        efs = np.repeat(efs[:, 2:3], 3, axis=1)
    nz = efs.flatten()
    nz = nz[nz > 0]
    upper = min(np.percentile(nz, upper_bound_percentile), ceil)
    logger.info(f"Upper bound of the event frame value during video writing: {upper}")
    efs = np.clip(efs, 0, upper) / upper
    return (np.moveaxis(efs, 1, -1) * 255).astype(np.uint8)


def write_event_frame_video(efs: np.ndarray, ef_video_path, fps, ceil, upper_bound_percentile=98, keep_polarity=True):
    """v2ce.py:241-280.  The frames are computed here; the mp4 container needs OpenCV (cv2.VideoWriter,
    'mp4v'), which the build image lacks -- then the same uint8 frames are written next to it as
    ``<stem>.npz`` (key ``event_frames``, RGB)."""
    frames = event_frame_images(efs, ceil, upper_bound_percentile, keep_polarity)
    try:
        import cv2
    except ImportError:
        alt = op.splitext(ef_video_path)[0] + ".npz"
        np.savez_compressed(alt, event_frames=frames, fps=np.float64(fps))
        logger.warning(f"OpenCV is not installed: event frames written to {alt} instead of {ef_video_path}")
        return alt
    H, W = frames.shape[1:3]
    video = cv2.VideoWriter(ef_video_path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (W, H))
    for fr in frames:
        video.write(cv2.cvtColor(fr, cv2.COLOR_RGB2BGR))
    video.release()
    logger.info(f"Event frame video written to {ef_video_path}")
    return ef_video_path


def run(frames: np.ndarray, model, infer_type="center", seq_len=16, width=346, height=260,
        batch_size=1, fps=30, stage2_batch_size=24, seed=0, rng="philox", device="cuda",
        stage2=None, event_frames: Optional[list] = None, comm=None, out_path: Optional[str] = None):
    """frames [N,H,W] uint8 -> event_stream (numpy structured array, v2ce.py:368) on rank 0.

    out_path (Philox draws only): write the reference's ``np.savez(out_path, event_stream=...)`` file WHILE the clip
    runs (``npz_stream.NpzStreamWriter`` behind ``pipeline.StreamingEventSink``: the host never holds the clip, the disk
    works under the GPU) and return the number of events on rank 0 instead of the array.

    Default (counter-based Philox draws): the per-batch pipeline of ``pipeline.run_clip`` (H2D,
    UNet + LDATI, D2H overlapped; under torch.distributed sharded over GPUs).  ``rng='torch'``
    reproduces the reference's structure instead -- the whole clip through the model, then LDATI in
    chunks of --stage2_batch_size with one dense torch.rand per chunk (LDATI.py:169-171) -- because
    that draw order depends on the chunking."""
    from . import pipeline
    comm = comm or vdist.default_comm(force=os.environ.get("V2CE_FORCE_DIST") == "1")
    world = comm.world
    if rng == "torch":
        if world > 1:
            raise NotImplementedError("rng='torch' replays the reference's single-process draw order; run it on one GPU")

        def clip():
            vox = glue.video_to_voxels(model, frames=frames, infer_type=infer_type, seq_len=seq_len,
                                       width=width, height=height, batch_size=batch_size, device=device)
            if event_frames is not None:
                from .pipeline import event_frame_sums
                del event_frames[:]
                event_frames.append((0, event_frame_sums(vox)))
            packed, _ = events_from_voxels(vox, fps, stage2_batch_size, seed, rng)
            return download_events(packed)
    else:
        def clip():
            if event_frames is not None:
                del event_frames[:]
            writer = None
            if out_path is not None and comm.rank == 0:     # (a range-guard rerun of the clip starts the file again)
                from .npz_stream import NpzStreamWriter
                writer = NpzStreamWriter(out_path, "event_stream", EVENT_DTYPE)
            out = pipeline.run_clip(frames, model, infer_type=infer_type, seq_len=seq_len, width=width, height=height,
                                    batch_size=batch_size, fps=fps, seed=seed, device=device, stage2=stage2,
                                    dtype=EVENT_DTYPE, comm=comm, writer=writer,
                                    event_frames=event_frames if world == 1 else None)
            return writer.count if writer is not None else out
    # the split-half convolutions report a dynamic-range bound; beyond its limit the clip is repeated on
    # the exact-f32 kernels (glue.run_guarded)
    return glue.run_guarded(model, clip, comm=comm)


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper()))
    world = int(os.environ.get("WORLD_SIZE", 1))
    device = args.device
    # V2CE_FORCE_DIST=1: take the torch.distributed code path (RCCL init, streamed gather, reductions) with a world
    # of one -- how a 1-GPU box exercises it (under torchrun --nproc-per-node 1)
    dist_on = world > 1 or os.environ.get("V2CE_FORCE_DIST") == "1"
    if dist_on:
        local = int(os.environ.get("LOCAL_RANK", 0))
        device = f"cuda:{local}"
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")
    elif str(device).startswith("cuda") and torch.device(device).index is not None:
        # the C ABI launches on the current device's streams: make --device cuda:N the current one
        torch.cuda.set_device(torch.device(device))
    sources = [args.image_folder, args.input_video_path, args.npy_frames, args.synthetic or None]
    assert sum(s is not None for s in sources) == 1, "specify exactly one frame source"
    if args.image_folder is not None:
        assert os.path.exists(args.image_folder), f"{args.image_folder} does not exist"
        name = Path(args.image_folder).name
        frames = read_image_folder(args.image_folder, args.max_frame_num)
    elif args.input_video_path is not None:
        assert os.path.exists(args.input_video_path), f"{args.input_video_path} does not exist"
        name = Path(args.input_video_path).stem
        frames = read_video(args.input_video_path, args.max_frame_num)
    elif args.npy_frames is not None:
        name = Path(args.npy_frames).stem
        frames = np.load(args.npy_frames)[:args.max_frame_num]
    else:
        from . import synth
        name = f"synthetic{args.synthetic}"
        frames = synth.synthetic_frames(args.synthetic, args.height, args.width)
    output_name = f"{name}-ceil_{args.ceil}-fps_{args.fps}" if args.out_name_suffix == "" else \
        f"{name}-ceil_{args.ceil}-fps_{args.fps}-{args.out_name_suffix}"
    os.makedirs(args.out_folder, exist_ok=True)
    if args.synthetic_weights is not None:
        from . import synth
        model = V2ce3d(precision=args.precision)
        model.load_state_dict(synth.make_state_dict(args.synthetic_weights))
        model = model.eval().to(device)
    else:
        model = get_trained_mode(args.model_path, device, args.precision)
    efs = [] if (args.write_event_frame_video and world == 1) else None
    if args.write_event_frame_video and world > 1:
        logger.warning("the event-frame video (v2ce.py:241-280) is written by single-process runs only: skipped")
    events_path = op.join(args.out_folder, f"{output_name}-events.npz")
    streaming = args.stream_events and args.rng == "philox"
    event_stream = run(frames, model, args.infer_type, args.seq_len, args.width, args.height,
                       args.batch_size, args.fps, args.stage2_batch_size, args.seed, args.rng, device,
                       event_frames=efs, out_path=events_path if streaming else None)
    if efs:
        ef = torch.cat([t for _, t in sorted(efs, key=lambda kv: kv[0])]).cpu().numpy()
        vis_color = "rgb" if args.vis_keep_polarity else "gray"
        os.makedirs(args.out_folder, exist_ok=True)
        write_event_frame_video(ef, op.join(args.out_folder, f"{args.infer_type}-{output_name}-pred_ef_{vis_color}.mp4"),
                                args.fps, args.ceil, args.upper_bound_percentile, args.vis_keep_polarity)
    if event_stream is not None:
        if streaming:                                       # the file is already there (npz_stream.NpzStreamWriter)
            logger.info(f"Generated event stream shape: , ({event_stream},)")
        else:
            logger.info(f"Generated event stream shape: , {event_stream.shape}")
            np.savez(events_path, event_stream=event_stream)
        print(events_path)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
