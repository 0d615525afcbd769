"""Deterministic synthetic inputs for parity tests, golden generation and bench.py.

No dataset and no checkpoint exists on the build or GPU box (SURVEY.md §8(c),(d)), so
every parity and throughput number is taken on

* a procedurally textured fronto-parallel plane seen by a ring of converging cameras
  (``synthetic_scene``): integer-hash value noise evaluated in float64 with + and *
  only, quantised to integer grey levels, so that the same tensors are regenerated
  bit-for-bit on any host;
* closed-form pseudo-random weights (``fill_state_dict``): a 32-bit integer mix of
  (tensor index, element index), scaled per tensor by its fan-in.

The loader tuple has the reference's shapes: ``images [1,V+1,3,H,W]`` float32 in
0..255, ``poses [1,V+1,4,4]`` world->camera, ``intrinsics [1,V+1,3,3]``, ``scale``
(reference: datasets/dtu.py returns exactly this tuple; inference.py:42).
"""
import math
import os

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """murmur3 finaliser on uint64 arrays holding 32-bit values (exact integer math)."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_uniform(n, seed):
    """n reproducible float64 values in [-1, 1)."""
    idx = np.arange(n, dtype=np.uint64)
    h = _mix32(idx * np.uint64(0x9E3779B1) + np.uint64(seed) * np.uint64(0x7F4A7C15))
    h = _mix32(h + np.uint64(0x165667B1))
    return h.astype(np.float64) / 2147483648.0 - 1.0


def _lattice(ix, iy, seed):
    h = _mix32(ix.astype(np.uint64) * np.uint64(0x27D4EB2F)
               + iy.astype(np.uint64) * np.uint64(0x165667B1)
               + np.uint64(seed) * np.uint64(0x9E3779B1))
    return _mix32(h).astype(np.float64) / 4294967296.0


def _value_noise(x, y, cell, seed):
    """Bilinear value noise with lattice spacing ``cell`` (float64, + and * only)."""
    gx = x / cell + 4096.0
    gy = y / cell + 4096.0
    x0 = np.floor(gx)
    y0 = np.floor(gy)
    fx = gx - x0
    fy = gy - y0
    ix = x0.astype(np.int64)
    iy = y0.astype(np.int64)
    a = _lattice(ix, iy, seed)
    b = _lattice(ix + 1, iy, seed)
    c = _lattice(ix, iy + 1, seed)
    d = _lattice(ix + 1, iy + 1, seed)
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def plane_texture(x, y, channel, seed):
    """Grey level in [0,255] of the textured plane at reference-pixel coords (x, y)."""
    t = np.zeros_like(x)
    amp_sum = 0.0
    for k, (cell, amp) in enumerate(((37.0, 0.30), (11.0, 0.30), (4.0, 0.25), (1.7, 0.15))):
        t = t + amp * _value_noise(x, y, cell, seed * 131 + channel * 17 + k)
        amp_sum += amp
    return 255.0 * t / amp_sum


def synthetic_scene(H, W, V, seed=0, depth=600.0, ring_deg=4.0, focal_factor=1.8):
    """1 reference + V source views of a plane at ``depth`` (DTU-like mm units).

    Source camera v sits on a circle of radius ``depth`` around the plane centre,
    ``ring_deg*ceil(v/2)`` degrees left/right (alternating) plus a small elevation,
    looking at the centre - so that every view overlaps the reference, as DTU rigs do.
    Returns (images, poses, intrinsics, scale) as CPU float32 tensors.
    """
    fx = focal_factor * W
    K = np.array([[fx, 0.0, W / 2.0], [0.0, fx, H / 2.0], [0.0, 0.0, 1.0]])
    Kinv = np.linalg.inv(K)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    images = np.zeros((V + 1, 3, H, W), dtype=np.float32)
    poses = np.zeros((V + 1, 4, 4), dtype=np.float64)
    centre = np.array([0.0, 0.0, depth])
    def render(v):                                         # views are independent: rendered by a small thread pool below
        if v == 0:
            R = np.eye(3)
            t = np.zeros(3)
        else:
            k = (v + 1) // 2
            yaw = math.radians(ring_deg * k) * (1.0 if v % 2 else -1.0)
            pitch = math.radians(0.6 * ring_deg * ((v % 3) - 1))
            cy, sy = math.cos(yaw), math.sin(yaw)
            cp, sp = math.cos(pitch), math.sin(pitch)
            Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
            Rx = np.array([[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]])
            R = Rx @ Ry
            # camera looks at the plane centre from distance ``depth``:  R*centre + t = (0,0,depth)
            t = centre - R @ centre
        poses[v, :3, :3] = R
        poses[v, :3, 3] = t
        poses[v, 3, 3] = 1.0
        # inverse map: source pixel -> point on the plane Z=depth (reference frame) -> reference pixel
        rays = Kinv @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
        Rt = R.T
        rr = Rt @ rays
        lam = (depth + (Rt @ t)[2]) / rr[2]
        X = rr * lam - (Rt @ t)[:, None]
        u = K @ X
        ux = (u[0] / u[2]).reshape(H, W)
        uy = (u[1] / u[2]).reshape(H, W)
        for c in range(3):
            g = plane_texture(ux, uy, c, seed)
            images[v, c] = np.floor(g + 0.5).astype(np.float32)

    workers = max(1, min(V + 1, (os.cpu_count() or 1) // 2, 16))
    if workers > 1 and H * W >= 256 * 256:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(render, range(V + 1)))
    else:
        for v in range(V + 1):
            render(v)
    intr = np.broadcast_to(K, (V + 1, 3, 3)).copy()
    return (torch.from_numpy(images)[None].contiguous(),
            torch.from_numpy(poses.astype(np.float32))[None].contiguous(),
            torch.from_numpy(intr.astype(np.float32))[None].contiguous(),
            torch.tensor([1.0], dtype=torch.float64))


def synthetic_depth_maps(H, W, V, seed=0, depth=600.0, ring_deg=4.0, focal_factor=1.8, noise=2e-3, outliers=0.08):
    """Per-view depth maps [V+1,H,W] (float32) of the ``synthetic_scene`` plane as an MVS network would deliver them: the
    analytic ray/plane depth of every view times (1 + smooth value noise of relative amplitude ``noise``), with a fraction
    ``outliers`` of 8x8 blocks pushed 3-12 % off - so that the geometric-consistency masks of the fusion step are neither
    empty nor full.  Same cameras as ``synthetic_scene`` (call it for images / poses / intrinsics)."""
    fx = focal_factor * W
    K = np.array([[fx, 0.0, W / 2.0], [0.0, fx, H / 2.0], [0.0, 0.0, 1.0]])
    Kinv = np.linalg.inv(K)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    centre = np.array([0.0, 0.0, depth])
    out = np.zeros((V + 1, H, W), dtype=np.float32)
    for v in range(V + 1):
        if v == 0:
            R, t = np.eye(3), np.zeros(3)
        else:
            k = (v + 1) // 2
            yaw = math.radians(ring_deg * k) * (1.0 if v % 2 else -1.0)
            pitch = math.radians(0.6 * ring_deg * ((v % 3) - 1))
            cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
            Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
            Rx = np.array([[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]])
            R = Rx @ Ry
            t = centre - R @ centre
        rays = Kinv @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
        Rt = R.T
        lam = ((depth + (Rt @ t)[2]) / (Rt @ rays)[2]).reshape(H, W)
        if noise == 0.0 and outliers == 0.0:                 # exact plane depths
            out[v] = lam.astype(np.float32)
            continue
        smooth = plane_texture(xs * 3.0, ys * 3.0, 0, seed + 100 + v) / 255.0 - 0.5          # value noise in [-0.5, 0.5]
        d = lam * (1.0 + 2.0 * noise * smooth)
        bx, by = (xs // 8).astype(np.int64), (ys // 8).astype(np.int64)
        hsh = _lattice(bx, by, seed * 97 + 7 + 1000 * v)                                     # block hashes in [0, 1)
        amp = 0.03 + 0.09 * _lattice(bx, by, seed * 97 + 9 + 1000 * v)
        sign = np.where(_lattice(bx, by, seed * 97 + 11 + 1000 * v) < 0.5, -1.0, 1.0)
        d = np.where(hsh < outliers, d * (1.0 + sign * amp), d)
        out[v] = d.astype(np.float32)
    return torch.from_numpy(out)


def fill_state_dict(state_dict, seed=0, gain=1.0):
    """Overwrite every tensor of a RAFT ``state_dict`` with closed-form values.

    Conv weights: uniform with std ``gain*sqrt(2/fan_in)`` (activations stay O(1)
    through the ReLU stacks); biases: uniform(-0.05, 0.05).  Tensors are visited in
    sorted-key order with the ``module.`` prefix stripped, so DataParallel-style and
    bare checkpoints get the same numbers.
    """
    out = {}
    keys = sorted(state_dict.keys(), key=lambda k: k[7:] if k.startswith("module.") else k)
    for tid, k in enumerate(keys):
        ref = state_dict[k]
        n = ref.numel()
        u = hash_uniform(n, seed * 1000 + tid + 1)
        if ref.dim() == 4:
            fan_in = ref.shape[1] * ref.shape[2] * ref.shape[3]
            std = gain * math.sqrt(2.0 / fan_in)
            # delta heads: scaled so that a random-weight network walks like a trained one -
            # stage 0 drifts up through the 0..0.0025 volume (~6e-5 +- 4e-5 per iteration),
            # stage 1 refines in steps of about one fine hypothesis (see the bias branch below)
            if k.endswith("delta0.2.weight"):
                std *= 0.003
            elif k.endswith("delta1.2.weight"):
                std *= 0.0008
            vals = u * (std * math.sqrt(3.0))
        elif k.endswith("delta0.2.bias"):
            vals = 0.006 + u * 0.0005
        elif k.endswith("delta1.2.bias"):
            vals = 0.0005 + u * 0.0001
        else:
            vals = u * 0.05
        out[k] = torch.from_numpy(vals.astype(np.float32)).reshape(ref.shape).clone()
    return out


def tensor_checksum(t):
    """Order-sensitive 64-bit checksum of a float32 tensor's bit pattern (fixture guard)."""
    a = np.ascontiguousarray(t.detach().cpu().numpy()).view(np.uint32).astype(np.uint64).ravel()
    w = (np.arange(a.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    return int(np.bitwise_xor.reduce(a * w) & np.uint64(0xFFFFFFFFFFFFFFFF))
