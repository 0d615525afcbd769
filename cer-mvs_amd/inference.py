"""Caller-side counterpart of the reference's inference driver (reference: inference.py:19-66,
utils/data_utils.py:58-78, utils/frame_utils.py:138-163): scale/crop of images + intrinsics,
model call, disparity -> depth, PFM writer.  Dataset readers are out of scope:
``inference`` takes any iterable yielding the reference's loader tuple
(images [1,N,3,H,W], poses [1,N,4,4], intrinsics [1,N,3,3], image_names, scale)."""
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from .raft import RAFT, SaturationError


def scale_operation(images, intrinsics, s):
    """images [N,3,H,W], intrinsics [N,3,3] -> bilinear (align_corners=True) resize to int(s*H) x int(s*W),
    fx, fy, cx, cy scaled (reference: utils/data_utils.py:58-66).  Out of place."""
    ht2, wd2 = int(s * images.shape[2]), int(s * images.shape[3])
    intrinsics = intrinsics.clone()
    intrinsics[:, 0] *= s
    intrinsics[:, 1] *= s
    return F.interpolate(images, [ht2, wd2], mode="bilinear", align_corners=True), intrinsics


def crop_operation(images, intrinsics, crop_h, crop_w):
    """Centre crop; principal point shifted (reference: utils/data_utils.py:69-78).  Out of place."""
    x0 = (images.shape[3] - crop_w) // 2
    y0 = (images.shape[2] - crop_h) // 2
    intrinsics = intrinsics.clone()
    intrinsics[:, 0, 2] -= x0
    intrinsics[:, 1, 2] -= y0
    return images[:, :, y0:y0 + crop_h, x0:x0 + crop_w], intrinsics


def disp_to_depth(res):
    """numpy [h,w] inverse depth -> float32 depth, 0 where disparity is 0 (reference: inference.py:57-58)."""
    with np.errstate(divide="ignore"):
        return np.where(res == 0, 0, 1 / res).astype(np.float32)


def write_pfm(file, image, scale=1):
    """float32 PFM as the reference writes it (utils/frame_utils.py:138-163): 'Pf' (H x W or H x W x 1) / 'PF' (H x W x 3), 'W H',
    the scale (negated = little endian), rows bottom-up."""
    if image.dtype != np.float32 or not (image.ndim == 2 or (image.ndim == 3 and image.shape[2] in (1, 3))):
        raise ValueError("write_pfm: float32 H x W, H x W x 1 or H x W x 3 expected")
    rows = np.ascontiguousarray(np.flipud(image).astype("<f4"))
    head = (b"PF\n" if (image.ndim == 3 and image.shape[2] == 3) else b"Pf\n") + b"%d %d\n" % (image.shape[1], image.shape[0]) + b"%f\n" % -scale
    Path(file).write_bytes(head + rows.tobytes())


def inference(test_loader, ckpt, output_folder, rescale=1, crop=None, do_report=False, write_min_depth=None,
              model=None, num_frames=None, streams=3):
    """Reference signature (inference.py:19-27) plus ``model`` (pre-built RAFT), ``num_frames`` (the reference reads
    test_loader.dataset.num_frames for the file name) and ``streams``: reference views kept in flight on the GPU
    (pipeline.DepthMapPipeline: 3 = +12-16 % depth maps per second at DTU size; 1 = the reference's one-at-a-time loop;
    ``do_report`` forces 1 so that the per-view time it prints means what it says).  Same files either way."""
    if model is None:
        model = RAFT(test_mode=True).cuda()
        if ckpt is not None:
            model.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=True)
    # the reference wraps its model in nn.DataParallel (inference.py:28-35): the pipeline works on the RAFT module itself
    core = model.module if hasattr(model, "module") and isinstance(getattr(model, "module"), RAFT) else model
    was_training = core.training
    core.eval()
    output_folder = Path(output_folder)
    (output_folder / "depths").mkdir(exist_ok=True, parents=True)
    written = []
    from .pipeline import DepthMapPipeline
    # Depth maps in flight: each extra stream is a REPLICA of the model (weights, packed weights, feature buffers, workspaces: ~3.7 GB
    # per replica at DTU size, ~32 GB at 3840x2160 x 15 views).  A sharded model (view_group) runs collectives inside its forward:
    # replicas issuing them from several streams in rank-dependent order could deadlock, and a process group is not deep-copyable - one
    # depth map at a time there (bench.py does the same).
    n_streams = 1 if (do_report or getattr(core, "view_group", None) is not None) else max(1, int(streams))
    pipe = DepthMapPipeline(core, streams=n_streams)
    pending = []

    def finish(entry):
        handle, name, nf = entry
        disp_est = pipe.result(handle)
        # saturation of a split-f16 operand is an error, not a silent clamp: under the lazy policy the flag of forward k is polled when
        # forward k + streams starts; nothing is written for a forward whose flag is already known to be set
        bits = pipe.poll_overflow()
        if bits:
            core._raise_overflow(bits)
        if do_report:
            torch.cuda.synchronize()
            print(f"per view time: {time.time() - tic[0]}")
        im = disp_to_depth(disp_est.cpu().numpy()[0, 0])
        path = output_folder / "depths" / f"{name}_scale{rescale}_nf{nf}.pfm"
        write_pfm(path, im)
        written.append(str(path))
        if write_min_depth is not None:
            wm = Path(write_min_depth)
            wm.mkdir(exist_ok=True)
            with open(wm / f"{name}.txt", "w") as f:
                f.write(f"{np.quantile(im[im > 0], 0.1) / 2}\n")

    tic = [0.0]
    n_flight = len(pipe)
    try:
      with torch.no_grad():
        for images, poses, intrinsics, image_names, scale in test_loader:
            poses = poses.cuda()
            images, intrinsics = scale_operation(images.squeeze(0), intrinsics.squeeze(0), rescale)
            if crop is not None:
                images, intrinsics = crop_operation(images, intrinsics, crop[0], crop[1])
            images = images.unsqueeze(0).cuda()
            intrinsics = intrinsics.unsqueeze(0).cuda()
            if do_report:
                torch.cuda.synchronize()
                tic[0] = time.time()
            name = image_names[0][0] if isinstance(image_names[0], (list, tuple)) else image_names[0]
            nf = num_frames if num_frames is not None else getattr(getattr(test_loader, "dataset", None), "num_frames", images.shape[1])
            pending.append((pipe.submit(images, poses, intrinsics, scale, do_report=do_report), name, nf))
            if len(pending) >= n_flight:
                finish(pending.pop(0))
        while pending:
            finish(pending.pop(0))
      pipe.check_overflow()                                 # (reads the flag: covers the last forwards, whose snapshots nobody polled)
    except SaturationError:
        # (ONE clean-up for both raise sites - the poll inside finish() and the final check; ADVICE r5: for THIS error only -
        # an out-of-memory or loader error must not delete depth maps that were written correctly.)  The files of the last
        # `streams` forwards were written before their flags could be read: remove them rather than leave saturated depth maps next to
        # good ones
        for path in written[-n_flight:]:
            try:
                Path(path).unlink()
            except OSError:
                pass
        raise
    finally:
        if was_training:
            core.train()
    return written
