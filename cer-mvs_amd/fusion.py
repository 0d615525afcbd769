"""Geometric-consistency filtering and fusion of the depth maps - the step after the inference path
(reference: fusion.py), on the MI355X through ``cer_geo_consistency_f32`` (csrc/fusion.hip).

Mirrors the reference's interface:
  * ``check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, thre1, thre2)`` - same arguments and
    return tuple as fusion.py:86-106 (batched over source views);
  * ``vote(...)`` - the fused form the fusion loop uses: one launch per reference view, only the vote mask, the averaged
    depth and the mask area leave the kernel;
  * ``fuse_depth_maps(depths, Ks, Es, pairs, glb)`` - the ten-round bisection of fusion.py:199-262 on device tensors;
  * ``fusion(data_loader, output_folder, suffix, glb, rescale)`` - the reference's driver: reads the ``depths/*.pfm`` that
    ``inference`` wrote, runs the loop, writes ``mask/<view>.png`` and ``result.ply``.
The camera algebra (3x3 / 4x4 inverses and products, a few hundred flops per view pair) stays on the host in fp32 torch,
in the reference's order; everything per pixel is in the kernel.  No CPU fallback: tensors must be CUDA tensors."""
import ctypes
import math
import os
import re

import numpy as np
import torch

from . import _lib as L

CAM_FLOATS = 60
COUNTERS = 64          # cer_mvs.h CER_GEO_COUNTERS: the mask area is accumulated over this many device counters


def compose_cams(K_ref, E_ref, K_src, E_src):
    """Per source view the six matrices of the reprojection chain, fp32, in the reference's evaluation order
    (fusion.py:50,55,58,71,74,78) -> CPU tensor [S, 60]."""
    K_ref, E_ref, K_src, E_src = (t.detach().to("cpu", torch.float32) for t in (K_ref, E_ref, K_src, E_src))
    if K_ref.dim() == 2:
        K_ref, E_ref = K_ref[None].expand(K_src.shape[0], 3, 3), E_ref[None].expand(K_src.shape[0], 4, 4)
    S = K_src.shape[0]
    A = torch.inverse(K_ref)
    Trs = torch.matmul(E_src, torch.inverse(E_ref))[:, :3]
    Ksi = torch.inverse(K_src)
    Tsr = torch.matmul(E_ref, torch.inverse(E_src))[:, :3]
    return torch.cat([A.reshape(S, 9), Trs.reshape(S, 12), K_src.reshape(S, 9), Ksi.reshape(S, 9), Tsr.reshape(S, 12),
                      K_ref.reshape(S, 9)], 1).contiguous()


def _launch(depth_ref, depth_src, cams, thre1, thre2, geo_mask=None, depth_est=None, count=None, literal=None):
    S, H, W = depth_src.shape
    lit = literal or {}
    p = lambda t, dt=torch.float32: L.dev_ptr(t, "tensor", dt)
    L.check(L.load().cer_geo_consistency_f32(
        p(depth_ref), p(depth_src), p(cams), S, H, W, float(thre1), float(thre2), p(geo_mask, torch.uint8), p(depth_est),
        p(count, torch.int32), p(lit.get("masks9"), torch.uint8), p(lit.get("drep")), p(lit.get("xs")), p(lit.get("ys")), p(lit.get("rel")),
        L.cur_stream()), "geo_consistency")


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, thre1=4.4, thre2=1430.0):
    """Same contract as the reference (fusion.py:86-106): all arguments batched over the S source views (the reference view's
    depth / cameras repeated S times, as the caller in fusion.py:218-220 does).  Returns
    (masks: list of 9 bool [S,H,W], mask, depth_reprojected (zero outside mask), x2d_src, y2d_src, relative_depth_diff)."""
    if not depth_ref.is_cuda:
        raise RuntimeError("check_geometric_consistency: depth maps must be CUDA tensors (no CPU fallback)")
    S, H, W = depth_src.shape
    dev = depth_ref.device
    if S > 1 and not (torch.equal(depth_ref[0], depth_ref[-1]) and torch.equal(intrinsics_ref[0], intrinsics_ref[-1])):
        raise ValueError("check_geometric_consistency: the reference view must be the same for every source view of a call")
    cams = compose_cams(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src).to(dev)
    lit = {"masks9": torch.empty(9, S, H, W, device=dev, dtype=torch.uint8)}
    for k in ("drep", "xs", "ys", "rel"):
        lit[k] = torch.empty(S, H, W, device=dev, dtype=torch.float32)
    _launch(depth_ref[0].float().contiguous(), depth_src.float().contiguous(), cams, thre1, thre2, literal=lit)
    masks = [lit["masks9"][i].bool() for i in range(9)]
    return masks, masks[-1], lit["drep"], lit["xs"], lit["ys"], lit["rel"]


def vote(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, thre1, thre2, cams=None, count=None):
    """One reference view [H,W] against its source views [S,H,W] (fusion.py:213-236), fused: returns (geo_mask uint8 [H,W],
    depth_est [H,W]); ``count`` (int32 [1], device) accumulates the mask area."""
    if not depth_ref.is_cuda:
        raise RuntimeError("vote: depth maps must be CUDA tensors (no CPU fallback)")
    S, H, W = depth_src.shape
    dev = depth_ref.device
    if cams is None:
        cams = compose_cams(K_ref, E_ref, K_src, E_src).to(dev)
    geo = torch.empty(H, W, device=dev, dtype=torch.uint8)
    est = torch.empty(H, W, device=dev, dtype=torch.float32)
    _launch(depth_ref, depth_src, cams, thre1, thre2, geo_mask=geo, depth_est=est, count=count)
    return geo, est


def fuse_depth_maps(depths, Ks, Es, pairs, glb=0.25, rounds=10):
    """The bisection of the threshold exponent on the mean mask area (fusion.py:199-262) on device tensors.
    depths [N,H,W] (CUDA), Ks [N,3,3], Es [N,4,4], pairs = [(ref index, [source indices])].
    Returns (masks uint8 [N,H,W], depth_est [N,H,W], exponent of the last round, [(exponent, mean area)] history)."""
    if not depths.is_cuda:
        raise RuntimeError("fuse_depth_maps: depth maps must be CUDA tensors (no CPU fallback)")
    dev = depths.device
    N, H, W = depths.shape
    depths = depths.float().contiguous()
    cams, srcs = [], []
    for ref, src in pairs:                                   # camera chains do not depend on the threshold: once
        cams.append(compose_cams(Ks[ref], Es[ref], Ks[src], Es[src]).to(dev))
        srcs.append(depths[src].contiguous())
    masks = torch.zeros(N, H, W, device=dev, dtype=torch.uint8)
    est = torch.zeros(N, H, W, device=dev, dtype=torch.float32)
    counts = torch.zeros(len(pairs), COUNTERS, device=dev, dtype=torch.int32)
    lo, hi, hist, thre = -2.0, 2.0, [], 0.0
    for _ in range(rounds):
        thre = (lo + hi) / 2
        counts.zero_()
        for k, (ref, _) in enumerate(pairs):
            _launch(depths[ref], srcs[k], cams[k], 10 ** thre * 4, 10 ** thre * 1300, geo_mask=masks[ref], depth_est=est[ref],
                    count=counts[k])
        cnt = counts.sum(1).cpu().numpy()                    # one synchronisation per round
        # the reference averages per-view float32 means, geo_mask.float().mean().item() (fusion.py:238, :272)
        mean = float(np.mean([float(np.float32(c) / np.float32(H * W)) for c in cnt]))
        hist.append((thre, mean))
        if mean >= glb:
            lo = thre
        else:
            hi = thre
    return masks, est, thre, hist


def backproject(depth, mask, K, E):
    """Masked pixels -> world points, numpy float64 like the reference (fusion.py:262-270)."""
    H, W = depth.shape
    x, y = np.meshgrid(np.arange(0, W), np.arange(0, H))
    x, y, d = x[mask], y[mask], depth[mask]
    cam = np.matmul(np.linalg.inv(K), np.vstack((x, y, np.ones_like(x))) * d)
    return np.matmul(np.linalg.inv(E), np.vstack((cam, np.ones_like(x))))[:3].transpose((1, 0))


def read_pfm(path):
    """Inverse of inference.write_pfm (reference: utils/frame_utils.py:31-66): float32 [H,W] (or [H,W,3]), top row first."""
    with open(path, "rb") as f:
        header = f.readline().rstrip()
        if header not in (b"PF", b"Pf"):
            raise ValueError(f"{path}: not a PFM file")
        m = re.match(rb"^(\d+)\s(\d+)\s*$", f.readline())
        if not m:
            raise ValueError(f"{path}: malformed PFM header")
        w, h = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
    shape = (h, w, 3) if header == b"PF" else (h, w)
    return np.flipud(data.reshape(shape)).astype(np.float32)


def write_ply(path, xyz, rgb):
    """Binary little-endian PLY with float x, y, z and uchar red, green, blue - what plyfile's PlyData([...]).write produces
    for the reference's vertex array (fusion.py:281-294)."""
    v = np.empty(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\n" f"element vertex {len(v)}\n"
              "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def _resize(img, h, w):
    """cv2.resize(..., INTER_LINEAR) of the reference (fusion.py:144,151): half-pixel-centre bilinear."""
    t = torch.from_numpy(np.ascontiguousarray(img)).float()
    t = t.permute(2, 0, 1)[None] if t.dim() == 3 else t[None, None]
    out = torch.nn.functional.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)[0]
    return (out.permute(1, 2, 0) if img.ndim == 3 else out[0]).numpy()


def fusion(data_loader, output_folder, suffix="", glb=0.25, rescale=1, device="cuda", write=True):
    """The reference's driver (fusion.py:110-297).  ``data_loader`` yields (images [1,n,3,H,W], extrinsics [1,n,4,4],
    intrinsics [1,n,3,3], image_names, _) per reference view; the estimated depth of view <name> is read from
    ``output_folder/depths/<name><suffix>.pfm``.  Writes ``mask/<index><suffix>.png`` and ``result.ply`` (``write``), returns
    {"masks", "depth_est", "xyz", "rgb", "threshold", "history"}."""
    output_folder = output_folder if hasattr(output_folder, "__truediv__") else __import__("pathlib").Path(output_folder)
    imgs, deps, Ks, Es, index_of, pair_names = [], [], [], [], {}, []
    for i, (images, extrinsics, intrinsics, image_names, _) in enumerate(data_loader):
        images = images.squeeze(0)
        E, K = extrinsics[0][0].clone().float(), intrinsics[0][0].clone().float()
        refid = image_names[0][0]
        index_of[refid] = i
        pair_names.append((refid, [x[0] for x in image_names[1:]]))
        img = images[0].permute(1, 2, 0).numpy() / 255.0
        dep = read_pfm(output_folder / "depths" / f"{refid}{suffix}.pfm")
        h, w = dep.shape
        if rescale != 1:
            dep = _resize(dep, int(h * rescale), int(w * rescale))
        scale = float(dep.shape[0]) / img.shape[0]
        flag = 0
        if dep.shape[1] / img.shape[1] > scale:
            scale = float(dep.shape[1]) / img.shape[1]
            flag = 1
        if scale != 1.0:
            img = _resize(img, int(round(img.shape[0] * scale)), int(round(img.shape[1] * scale)))
        if flag == 0:
            index = int(math.ceil((img.shape[1] - dep.shape[1]) / 2))
            img = img[:, index:dep.shape[1] + index, :]
        else:
            index = int(math.ceil((img.shape[0] - dep.shape[0]) / 2))
            img = img[index:img.shape[0] - index, :, :]
        K[:2, :] *= scale                                    # modify_camera_parameters (fusion.py:24-30)
        K[0 if flag == 0 else 1, 2] -= index
        if i > 0 and (img.shape != imgs[0].shape or dep.shape != deps[0].shape):      # fusion.py:176-189: crop / zero-pad to view 0
            ih, iw = imgs[0].shape[:2]
            pi = np.zeros_like(imgs[0]); pi[:min(ih, img.shape[0]), :min(iw, img.shape[1])] = img[:ih, :iw]
            dh, dw = deps[0].shape
            pd = np.zeros_like(deps[0]); pd[:min(dh, dep.shape[0]), :min(dw, dep.shape[1])] = dep[:dh, :dw]
            img, dep = pi, pd
        imgs.append(img); deps.append(dep); Ks.append(K); Es.append(E)
    dev = torch.device(device)
    depths = torch.from_numpy(np.stack(deps)).float().to(dev)
    Ks, Es = torch.stack(Ks), torch.stack(Es)
    pairs = [(index_of[r], [index_of[s] for s in ss]) for r, ss in pair_names]
    masks, est, thre, hist = fuse_depth_maps(depths, Ks, Es, pairs, glb=glb)
    masks_np, est_np = masks.cpu().numpy().astype(bool), est.cpu().numpy()
    xyz, rgb = [], []
    for ref, _ in pairs:
        xyz.append(backproject(est_np[ref], masks_np[ref], Ks[ref].numpy(), Es[ref].numpy()))
        rgb.append((imgs[ref][masks_np[ref]] * 255).astype(np.uint8))
    xyz, rgb = np.concatenate(xyz, 0), np.concatenate(rgb, 0)
    if write:
        os.makedirs(os.path.join(output_folder, "mask"), exist_ok=True)
        try:
            from PIL import Image
            for ref, _ in pairs:
                Image.fromarray(masks_np[ref].astype(np.uint8) * 255).save(str(output_folder / "mask" / f"{ref}{suffix}.png"))
        except ImportError:                                  # no PNG encoder in this Python: keep the arrays
            for ref, _ in pairs:
                np.save(str(output_folder / "mask" / f"{ref}{suffix}.npy"), masks_np[ref])
        write_ply(os.path.join(output_folder, "result.ply"), xyz.astype(np.float32), rgb)
    return {"masks": masks_np, "depth_est": est_np, "xyz": xyz, "rgb": rgb, "threshold": thre, "history": hist}
