"""Multi-resolution merge of the depth maps of two inference passes (reference: multires.py:16-40): for every view the scale-1
map is resized to the scale-2 map's size and the scale-2 depth is kept where the two agree to a relative threshold,

    im = where(|im1 - im2| < th * im1, im2, im1),

optionally down-sampled, and written as ``{name}{suffix1}{suffix2}_th{th}.pfm`` next to the inputs.  Same arguments, file names
and outputs as the reference; the resize + select runs in one HIP kernel (cer_multires_merge_f32: cv2.resize(INTER_LINEAR)
semantics on float32), the optional down-sampling in cer_resize_linear_f32."""
import os
from pathlib import Path

import numpy as np
import torch

from . import _lib as L
from .fusion import read_pfm
from .inference import write_pfm


def merge(im1, im2, th=0.02, down_sample=1, device=None):
    """im1 [h1,w1], im2 [h2,w2] (numpy float32 or tensors) -> merged float32 tensor [h2 // down_sample, w2 // down_sample] on the GPU."""
    dev = torch.device(device) if device is not None else (im2.device if isinstance(im2, torch.Tensor) and im2.is_cuda else torch.device("cuda"))
    a = torch.as_tensor(np.ascontiguousarray(im1) if isinstance(im1, np.ndarray) else im1, dtype=torch.float32).to(dev).contiguous()
    b = torch.as_tensor(np.ascontiguousarray(im2) if isinstance(im2, np.ndarray) else im2, dtype=torch.float32).to(dev).contiguous()
    if a.dim() != 2 or b.dim() != 2:
        raise RuntimeError("multires.merge: depth maps must be 2-D")
    lib = L.load()
    out = torch.empty_like(b)
    L.check(lib.cer_multires_merge_f32(L.dev_ptr(a, "im1"), a.shape[0], a.shape[1], L.dev_ptr(b, "im2"), b.shape[0], b.shape[1], float(th),
                                       L.dev_ptr(out, "out"), L.cur_stream()), "multires_merge")
    if down_sample != 1:
        ho, wo = b.shape[0] // down_sample, b.shape[1] // down_sample          # (reference: tuple(np.array(im.shape[::-1]) // down_sample))
        small = torch.empty(ho, wo, device=dev, dtype=torch.float32)
        L.check(lib.cer_resize_linear_f32(L.dev_ptr(out, "src"), b.shape[0], b.shape[1], L.dev_ptr(small, "dst"), ho, wo, L.cur_stream()), "resize_linear")
        out = small
    return out


def multires(output_folder, suffix1="", suffix2="", th=0.02, down_sample=1, visualize=False):
    """Reference signature and file contract (multires.py:16-40): reads ``depths/{name}_scale1{suffix1}.pfm`` and
    ``depths/{name}_scale2{suffix2}.pfm``, writes ``depths/{name}{suffix1}{suffix2}_th{th}.pfm`` (and, with ``visualize``, the
    inverse-depth picture ``depths/{name}.png``)."""
    output_folder = Path(output_folder)
    names = os.listdir(output_folder / "depths")
    names = sorted([name.split("_scale1")[0] for name in names if "_scale1" in name])
    written = []
    for name in names:
        output = output_folder / "depths" / f"{name}{suffix1}{suffix2}_th{th}.pfm"
        im1 = read_pfm(output_folder / "depths" / f"{name}_scale1{suffix1}.pfm")
        im2 = read_pfm(output_folder / "depths" / f"{name}_scale2{suffix2}.pfm")
        im = merge(im1, im2, th, down_sample).cpu().numpy()
        write_pfm(output, im)
        written.append(output)
        if visualize:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            with np.errstate(divide="ignore", invalid="ignore"):
                d = 1 / im
            d[np.isnan(d)] = 0
            d = np.minimum(np.maximum(d, 0), 5 * np.median(d))
            plt.figure(figsize=(20, 20))
            plt.imshow(d)
            plt.savefig(output_folder / "depths" / f"{name}.png")
            plt.close()
    return written
