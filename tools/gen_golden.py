#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (read-only at
/root/reference) on CPU under dependency shims.  Runs in the build container only;
the reference never travels, the .npz vectors (inputs' checksums + expected outputs) do.

Shims - none of them restates reference logic except (iv):
  (i)   gin            -> pass-through ``configurable`` decorator
  (ii)  fastcore.all   -> ``store_attr`` copying the caller's ctor args onto self
  (iii) opt_einsum     -> ``contract = torch.einsum``
  (iv)  alt_cuda_corr  -> CPU gather form of correlation_kernel.cu:59-116 (floor, 4 corners,
                          zero outside, (1-dy|dy)(1-dx|dx) weights).  Deliberately NOT the
                          grid_sample form the oracle uses, so the two restatements check
                          each other.
  (v)   cv2            -> empty module (only needed to import utils/frame_utils.py)
  (vi)  Tensor.cuda / Module.cuda -> identity (no GPU here)

usage: python tools/gen_golden.py [--only NAME]
"""
import argparse
import inspect
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def install_shims():
    gin = types.ModuleType("gin")

    def configurable(*dargs, **dkw):
        if len(dargs) == 1 and callable(dargs[0]) and not dkw:
            return dargs[0]
        return lambda obj: obj

    gin.configurable = configurable
    sys.modules["gin"] = gin

    fastcore = types.ModuleType("fastcore")
    fc_all = types.ModuleType("fastcore.all")

    def store_attr():
        frame = inspect.currentframe().f_back
        loc = frame.f_locals
        self = loc["self"]
        for k, v in loc.items():
            if k not in ("self", "__class__"):
                setattr(self, k, v)

    fc_all.store_attr = store_attr
    fastcore.all = fc_all
    sys.modules["fastcore"] = fastcore
    sys.modules["fastcore.all"] = fc_all

    oe = types.ModuleType("opt_einsum")
    oe.contract = torch.einsum
    sys.modules["opt_einsum"] = oe

    sys.modules["cv2"] = types.ModuleType("cv2")

    acc = types.ModuleType("alt_cuda_corr")

    def forward(fmap1, fmap2, coords, radius):
        assert radius == 0
        B, H1, W1, C = fmap1.shape
        _, H2, W2, _ = fmap2.shape
        N = coords.shape[1]
        out = torch.zeros(B, N, H1, W1)
        f2 = fmap2.reshape(B, H2 * W2, C)
        for n in range(N):
            x = coords[:, n, :, :, 0]
            y = coords[:, n, :, :, 1]
            x0 = torch.floor(x)
            y0 = torch.floor(y)
            dx = x - x0
            dy = y - y0
            acc_n = torch.zeros(B, H1, W1)
            for iy in (0, 1):
                for ix in (0, 1):
                    h2 = y0.to(torch.int64) + iy
                    w2 = x0.to(torch.int64) + ix
                    inb = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                    idx = (h2.clamp(0, H2 - 1) * W2 + w2.clamp(0, W2 - 1)).reshape(B, H1 * W1)
                    g = torch.gather(f2, 1, idx[..., None].expand(B, H1 * W1, C)).reshape(B, H1, W1, C)
                    s = (g * fmap1).sum(-1) * inb
                    wy = dy if iy else (1 - dy)
                    wx = dx if ix else (1 - dx)
                    acc_n = acc_n + s * wy * wx
            out[:, n] = acc_n
        return [out[:, :, None]]

    def backward(fmap1, fmap2, coords, corr_grad, radius):
        # correlation_kernel.cu:122-256 accumulates d corr / d fmap1 and d corr / d fmap2 and leaves coords_grad at its zero
        # fill (:307); here: autograd through the gather form above
        a, b = fmap1.detach().clone().requires_grad_(True), fmap2.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            out, = forward(a, b, coords.detach(), radius)
            (out * corr_grad).sum().backward()
        return [a.grad, b.grad, torch.zeros_like(coords)]

    acc.forward = forward
    acc.backward = backward
    sys.modules["alt_cuda_corr"] = acc

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def hashed(shape, seed, lo=-1.0, hi=1.0):
    from cer_mvs_amd.synthetic import hash_uniform
    n = int(np.prod(shape))
    u = hash_uniform(n, seed)
    return torch.from_numpy((lo + (u + 1) * 0.5 * (hi - lo)).astype(np.float32)).reshape(shape)


def feature_geometry(h1, w1, V):
    """poses + feature-resolution intrinsics of the synthetic rig (what RAFT.forward hands CorrBlock)."""
    from cer_mvs_amd.synthetic import synthetic_scene
    _, poses, intr, _ = synthetic_scene(8, 8, V, seed=1)          # images unused
    intr = intr.clone()
    fx = 1.8 * (4 * w1)
    intr[:, :, 0, 0] = fx / 4
    intr[:, :, 1, 1] = fx / 4
    intr[:, :, 0, 2] = (4 * w1) / 2 / 4
    intr[:, :, 1, 2] = (4 * h1) / 2 / 4
    return poses, intr


def gen_corrblock():
    from core.corr import CorrBlock
    h1, w1, V, C = 16, 24, 3, 64
    fmaps = hashed((1, V + 1, C, h1, w1), 11, -2.0, 2.0)
    poses, intr = feature_geometry(h1, w1, V)
    ii = torch.zeros(V, dtype=torch.long)
    jj = torch.arange(1, V + 1)
    out = {"h1": h1, "w1": w1, "V": V, "poses": poses.numpy(), "intrinsics": intr.numpy()}
    for stage, (D, N, shift) in enumerate(((64, 64, True), (44, 320, False))):
        incre = 0.0025 / N
        if shift:
            disp_in = hashed((1, 1, h1, w1), 21, -0.0005, 0.0022)
            disp_in = torch.where(disp_in < 0, torch.zeros_like(disp_in), disp_in)   # mix of below / above the shift limit
            zinv = hashed((1, 1, h1, w1), 22, -0.0002, 0.0029)
        else:
            disp_in = hashed((1, 1, h1, w1), 23, 0.0012, 0.0021)
            zinv = disp_in + hashed((1, 1, h1, w1), 24, -30.0, 30.0) * incre
        cb = CorrBlock(fmaps, poses, intr, ii, jj, nIncre=D, incre=incre, disps_input=disp_in, shift=shift,
                       num_levels=3, radius=5, test_mode=True, do_report=False)
        feats = cb(zinv[:, [0] * V])
        out[f"disp_in{stage}"] = disp_in.numpy()
        out[f"zinv{stage}"] = zinv.numpy()
        out[f"origin{stage}"] = cb.disps_origin.reshape(h1, w1).numpy()
        for lv, t in enumerate(cb.corr_pyramid):
            out[f"pyr{stage}_{lv}"] = t.reshape(V, h1 * w1, -1).numpy()
        out[f"feats{stage}"] = feats.numpy()
    np.savez_compressed(os.path.join(OUT, "corrblock.npz"), **out)
    print("corrblock.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


def ref_model(cascade, seed):
    from core.raft import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict
    model = RAFT(cascade=cascade, test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=seed), strict=True)
    model.eval()
    return model


def gen_update():
    h1, w1, V = 16, 24, 3
    model = ref_model([(64, 64, 1), (-1, 320, 1)], seed=3)
    net = torch.tanh(hashed((1, 1, 64, h1, w1), 31, -2, 2))
    inp = torch.relu(hashed((1, 1, 64, h1, w1), 32, -1, 2))
    disp = hashed((1, 1, h1, w1), 33, 0.0, 0.0025)
    corr = hashed((1, V, 33, h1, w1), 34, -1.5, 3.0)
    out = {"h1": h1, "w1": w1, "V": V, "weight_seed": 3}
    with torch.no_grad():
        for stage in (0, 1):
            n2, delta = model.update_block(net.clone(), inp.clone(), disp.clone(), corr.clone(), stage)
            out[f"net{stage}"] = n2.numpy()
            out[f"delta{stage}"] = delta.numpy()
    np.savez_compressed(os.path.join(OUT, "update.npz"), **out)
    print("update.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_e2e(name, H, W, V, cascade, seed):
    from cer_mvs_amd.synthetic import synthetic_scene, tensor_checksum
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=seed)
    model = ref_model([tuple(c) for c in cascade], seed=seed + 5)
    with torch.no_grad():
        disp = model(images.clone(), poses.clone(), intr.clone(), scale=scale.clone())
    out = {"H": H, "W": W, "V": V, "cascade": np.array(cascade), "scene_seed": seed, "weight_seed": seed + 5,
           "images_checksum": np.uint64(tensor_checksum(images)), "poses": poses.numpy(),
           "intrinsics": intr.numpy(), "disp": disp.numpy()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "disp", disp.shape, float(disp.min()), float(disp.mean()), float(disp.max()))


TRAIN_GRADS = ("fnet.conv1.weight", "fnet.conv2.weight", "cnet.conv2.weight", "update_block.gru.convz.weight",
               "update_block.corr_encoder.0.weight", "update_block.delta0.0.weight", "update_block.delta1.2.weight")


def gen_train_tiny():
    """The training row pinned to the reference itself: RAFT(test_mode=False) (core/raft.py:62-64,103,109) -> the list of
    predictions, loss.sequence_loss (loss.py:5-41) -> loss + metrics, and the gradients of seven parameters (through
    DirectCorr.backward, core/corr.py:19-25, with the shim's autograd form of the CUDA backward)."""
    from core.raft import RAFT
    from loss import sequence_loss
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene, tensor_checksum
    H, W, V, seed = 64, 96, 3, 2
    cascade = [(64, 64, 3), (-1, 320, 3)]
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=seed)
    model = RAFT(cascade=cascade, test_mode=False)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=seed + 5), strict=True)
    model.train()
    preds = model(images.clone(), poses.clone(), intr.clone(), scale=scale.clone().float())
    import torch.nn.functional as F
    gt = F.interpolate(preds[-1].detach() * 1.1 + 1e-4, [H, W], mode="bilinear", align_corners=True).clamp_min(1e-4)
    gt[:, :, :9, :] = 0.0                                          # invalid rows
    gt[:, :, 20:31, 40:57] = 0.0                                   # an invalid block
    loss, metrics = sequence_loss(list(preds), gt, gradual_weight=0.3)
    loss.backward()
    out = {"H": H, "W": W, "V": V, "cascade": np.array(cascade), "scene_seed": seed, "weight_seed": seed + 5,
           "images_checksum": np.uint64(tensor_checksum(images)), "gt": gt.numpy(), "gradual_weight": 0.3,
           "predictions": torch.stack([p_.detach() for p_ in preds]).numpy(), "loss": np.float64(loss.item()),
           "metrics": np.array([metrics[k] for k in ("mean_depth_error", "less3", "less10", "less25")], dtype=np.float64)}
    params = dict(model.named_parameters())
    for name in TRAIN_GRADS:
        g = params[name].grad.detach().reshape(-1)
        out["gradsum_" + name] = np.float64(g.double().abs().sum().item())
        out["grad_" + name] = (g if g.numel() <= 20000 else g[::7]).numpy()       # large tensors: every 7th element + the L1 norm
    np.savez_compressed(os.path.join(OUT, "train_tiny.npz"), **out)
    print("train_tiny.npz loss", float(loss), metrics, {k: v.shape for k, v in out.items() if k.startswith("grad_")})


def gen_e2e_lr():
    """encoder_type="LR" (core/extractor.py:87-90,151; core/raft.py:38: features at 1/8 resolution), end to end at a tiny size."""
    from core.raft import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene, tensor_checksum
    H, W, V, seed = 64, 96, 3, 3
    cascade = [(64, 64, 2), (-1, 320, 2)]
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=seed)
    model = RAFT(cascade=cascade, encoder_type="LR", test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=seed + 5), strict=True)
    model.eval()
    with torch.no_grad():
        disp = model(images.clone(), poses.clone(), intr.clone(), scale=scale.clone())
    out = {"H": H, "W": W, "V": V, "cascade": np.array(cascade), "scene_seed": seed, "weight_seed": seed + 5,
           "images_checksum": np.uint64(tensor_checksum(images)), "disp": disp.numpy()}
    np.savez_compressed(os.path.join(OUT, "e2e_lr.npz"), **out)
    print("e2e_lr", disp.shape, float(disp.min()), float(disp.mean()), float(disp.max()))


def gen_caller():
    from utils.data_utils import crop_operation, scale_operation
    from utils.frame_utils import write_pfm
    images = hashed((3, 3, 20, 28), 41, 0, 255)
    intr = torch.tensor([[50.0, 0, 14.0], [0, 50.0, 10.0], [0, 0, 1]]).repeat(3, 1, 1)
    out = {"images": images.numpy(), "intrinsics": intr.numpy()}
    im2, k2 = scale_operation(images.clone(), intr.clone(), 1.5)
    out["scaled_images"] = im2.numpy()
    out["scaled_intrinsics"] = k2.numpy()
    im3, k3 = crop_operation(im2.clone(), k2.clone(), 24, 32)
    out["cropped_images"] = im3.numpy()
    out["cropped_intrinsics"] = k3.numpy()
    disp = hashed((9, 13), 42, 0.0, 0.003)
    disp[2, 3] = 0.0
    res = disp.numpy()
    depth = np.where(res == 0, 0, 1 / res).astype(np.float32)      # inference.py:57-58, quoted as the expected transform
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "d.pfm")
        write_pfm(p, depth)
        out["pfm"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
    out["disp"] = res
    out["depth"] = depth
    np.savez_compressed(os.path.join(OUT, "caller.npz"), **out)
    print("caller.npz ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    torch.manual_seed(0)
    jobs = {
        "corrblock": gen_corrblock,
        "update": gen_update,
        "e2e_tiny": lambda: gen_e2e("e2e_tiny", 64, 96, 3, [(64, 64, 3), (-1, 320, 3)], seed=2),
        "e2e_cfg1": lambda: gen_e2e("e2e_cfg1", 480, 640, 2, [(64, 64, 2), (-1, 320, 2)], seed=0),
        "caller": gen_caller,
        "train_tiny": gen_train_tiny,
        "e2e_lr": gen_e2e_lr,
    }
    # BASELINE.json configs[1] - the bench workload itself (1600x1184, 10 source views, 32 GRU iterations; same scene and weight
    # seeds as bench.py).  Minutes of CPU time, so only on request: --only e2e_cfg2
    big = {"e2e_cfg2": lambda: gen_e2e("e2e_cfg2", 1184, 1600, 10, [(64, 64, 16), (-1, 320, 16)], seed=0),
           # BASELINE.json configs[4] (BlendedMVS 2048x1536, 7 source views) and configs[2] (Tanks&Temples 3840x2160, 15 source
           # views) at the cascades of bench.py's workloads of the same names
           "e2e_blended": lambda: gen_e2e("e2e_blended", 1536, 2048, 7, [(64, 64, 8), (-1, 320, 8)], seed=0),
           "e2e_tnt": lambda: gen_e2e("e2e_tnt", 2160, 3840, 15, [(64, 64, 8), (-1, 320, 8)], seed=0)}
    for name, fn in jobs.items():
        if args.only in (None, name):
            fn()
    if args.only in big:
        big[args.only]()


if __name__ == "__main__":
    main()
