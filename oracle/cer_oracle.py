"""ORACLE (test infrastructure - NOT the product path).

A CPU, fp32, functional restatement of the CER-MVS depth-inference hot path, written
from a reading of the reference (file:line cited per function; the reference lives at
/root/reference and never ships).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``cer-mvs_amd`` never does and fails loudly when its HIP library is missing.

Pinning: the reference has no tests and no golden vectors of its own, so
this restatement is pinned against tensors captured from the *reference's own Python*
imported under dependency shims in the build container (tools/gen_golden.py ->
tests/golden/*.npz; checked by tests/test_oracle_golden.py).  The one piece of the
reference that cannot be executed here is the CUDA kernel
alt_cuda_corr/correlation_kernel.cu:18-119 (no CUDA, no GPU in the container): its
arithmetic is pinned by reading only, restated twice (``alt_corr_forward`` below via
grid_sample, and literally, loop for loop, in oracle/cer_oracle.c) and the two are
checked against each other.  Parity for that kernel is therefore "pinned by
restatement", everything else "pinned by reference capture".

All tensors are torch CPU float32; B (batch) is 1 throughout, as at inference
(inference.py:42, batch_size=1).
"""
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- encoders


def _norm(x, kind):
    # core/extractor.py:28-31,75-76 - InstanceNorm2d defaults: no affine, no running stats, eps 1e-5
    return F.instance_norm(x, eps=1e-5) if kind == "instance" else x


def _res_block(x, sd, p, kind, stride):
    # core/extractor.py:49-57
    y = F.relu(_norm(F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1), kind))
    y = F.relu(_norm(F.conv2d(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), kind))
    if stride != 1:
        x = _norm(F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride), kind)
    return F.relu(x + y)


def encoder(x, sd, prefix, kind):
    """BasicEncoder.forward, type "HR" (core/extractor.py:143-155): [N,3,H,W] -> [N,Cout,H/4,W/4]."""
    x = F.relu(_norm(F.conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], stride=2, padding=3), kind))
    x = _res_block(x, sd, prefix + "layer1.0.", kind, 1)
    x = _res_block(x, sd, prefix + "layer1.1.", kind, 1)
    x = _res_block(x, sd, prefix + "layer2.0.", kind, 2)
    x = _res_block(x, sd, prefix + "layer2.1.", kind, 1)
    return F.conv2d(x, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])


# ------------------------------------------------------------------- projective geometry


def pij_matrices(poses, intrinsics, ii, jj):
    """Pij = K_j P_j P_i^-1 K_i^-1 (utils/projective_ops.py:18-23). poses [N,4,4], intrinsics [N,3,3]
    (already divided by the feature stride) -> [len(jj),4,4]."""
    Ks = torch.zeros_like(poses)
    Ks[:, :3, :3] = intrinsics
    Ks[:, 3, 3] = 1.0
    return Ks[jj] @ poses[jj] @ torch.inverse(poses[ii]) @ torch.inverse(Ks[ii])


def project(Pij, disps):
    """x1 = Pij (x, y, 1, d); (x1/x1_z)[..., :2] clamped to +-1e4
    (utils/projective_ops.py:5-13,26-28; core/corr.py:87-88).  Pij [4,4], disps [D,h,w] -> [D,h,w,2]."""
    D, h, w = disps.shape
    y, x = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    x0 = torch.stack([x.expand(D, h, w), y.expand(D, h, w), torch.ones(D, h, w), disps], -1)
    x1 = torch.einsum("kh,dyxh->dyxk", Pij, x0)
    x1 = x1 / x1[..., 2:3]
    return x1[..., :2].clamp(min=-1e4, max=1e4)


# ----------------------------------------------------------------------- correlation ops


def alt_corr_forward(fmap1, fmap2, coords, radius=0):
    """alt_cuda_corr.forward for radius 0 (alt_cuda_corr/correlation_kernel.cu:59-116, correlation.cpp:23-33):
    corr[b,n,0,h,w] = sum_c fmap1[b,h,w,c] * bilerp(fmap2[b], coords[b,n,h,w]); texels outside fmap2 read 0.
    fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2] (x,y in fmap2 pixels) -> [B,N,1,H1,W1]."""
    assert radius == 0, "the hot path only ever calls radius=0 (core/corr.py:16)"
    B, H2, W2, C = fmap2.shape
    img = fmap2.permute(0, 3, 1, 2)
    out = []
    for n in range(coords.shape[1]):
        g = coords[:, n]
        gx = 2 * g[..., 0] / (W2 - 1) - 1
        gy = 2 * g[..., 1] / (H2 - 1) - 1
        s = F.grid_sample(img, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="zeros", align_corners=True)
        out.append((s.permute(0, 2, 3, 1) * fmap1).sum(-1))
    return torch.stack(out, 1)[:, :, None]


def hypothesis_origin(disp_in, D, incre, shift):
    """core/corr.py:59-62 - stage-0 shift keeps every hypothesis >= 0."""
    if shift:
        lim = torch.tensor((D // 2) * incre, dtype=torch.float32)
        return torch.where(disp_in < (D // 2) * incre, lim, disp_in)
    return disp_in.clone()


def cost_volume(fmaps, poses, intrinsics, D, incre, disp_in, shift):
    """CorrBlock.__init__ up to the per-view volume (core/corr.py:46-91, 28-43).
    fmaps [V+1,C,h,w]; poses [V+1,4,4]; intrinsics [V+1,3,3] (feature-res); disp_in [h,w].
    Returns (vol [V,P,D], origin [h,w])."""
    N, C, h, w = fmaps.shape
    V = N - 1
    origin = hypothesis_origin(disp_in, D, incre, shift)
    disps = ((torch.arange(D) - D // 2) * incre).to(torch.float32).view(D, 1, 1) + origin[None]
    nhwc = fmaps.permute(0, 2, 3, 1)
    f1 = (nhwc[0:1] / 8.0).contiguous()
    vols = []
    for v in range(1, N):
        Pij = pij_matrices(poses, intrinsics, [0], [v])[0]
        xy = project(Pij, disps)                         # [D,h,w,2]
        f2 = (nhwc[v:v + 1] / 8.0).contiguous()
        c = alt_corr_forward(f1, f2, xy[None])           # [1,D,1,h,w]
        vols.append(c[0, :, 0].permute(1, 2, 0).reshape(h * w, D))
    return torch.stack(vols, 0), origin


def pyramid(vol, num_levels):
    """core/corr.py:94-97 - pairwise mean along D (avg_pool [1,2], floor)."""
    levels = [vol]
    for _ in range(num_levels - 1):
        v = levels[-1]
        lead = v.shape[:-1]
        v = F.avg_pool2d(v.reshape(-1, 1, 1, v.shape[-1]), [1, 2], stride=[1, 2])
        levels.append(v.reshape(*lead, -1))
    return levels


def sample_row(row, x):
    """bilinear_sampler1 on a 1 x W row image (utils/bilinear_sampler.py:6-25): pixel coord -> normalised
    -> F.grid_sample(align_corners=True, zeros).  row [N,W], x [N] -> [N]."""
    N, W = row.shape
    xg = 2 * x / (W - 1) - 1
    grid = torch.stack([xg, torch.zeros_like(xg)], -1).view(N, 1, 1, 2)
    return F.grid_sample(row.view(N, 1, 1, W), grid, mode="bilinear", padding_mode="zeros", align_corners=True).view(N)


def lookup(levels, origin, disp, D, incre, radius):
    """CorrBlock.__call__ (core/corr.py:102-143): levels list of [V,P,W_i]; origin, disp [h,w]
    -> [V, L*(2r+1), h, w]; channel = level*(2r+1) + (dx+r)."""
    V, P, _ = levels[0].shape
    h, w = disp.shape
    coords = torch.clamp_min((disp - origin) / incre + D // 2, 0.0).reshape(1, P).expand(V, P).reshape(-1)
    out = []
    for i, lv in enumerate(levels):
        rows = lv.reshape(V * P, -1)
        for dx in range(-radius, radius + 1):
            out.append(sample_row(rows, coords / 2 ** i + dx).view(V, P))
    return torch.stack(out, 1).view(V, -1, h, w)


# --------------------------------------------------------------------------- update block


def disp_features(disp, k=7):
    """UpdateBlock.disp_encoder (core/update.py:80-85): unfold k x k (zero pad) minus centre. [1,1,h,w] -> [1,49,h,w]."""
    _, _, h, w = disp.shape
    u = F.unfold(disp, [k, k], padding=k // 2).view(1, k * k, h, w)
    return u - disp


def update_block(sd, net, inp, disp, corr_frames, stage, prefix="update_block.", aggregation=("mean",)):
    """UpdateBlock.forward + ConvGRU.forward, defaults aggregation=["mean"], shared corr_encoder/gru,
    per-stage delta (core/update.py:87-120, 17-25).  net, inp [1,64,h,w]; disp [1,1,h,w];
    corr_frames [V,33,h,w] -> (net [1,64,h,w], delta [1,1,h,w])."""
    g = lambda n: sd[prefix + n]
    d = 100 * disp_features(disp)
    parts = []                                    # core/update.py:101-110: mean / max / std over views, stacked per channel
    if "mean" in aggregation:
        parts.append(corr_frames.mean(0))
    if "max" in aggregation:
        parts.append(corr_frames.max(0).values)
    if "std" in aggregation:
        parts.append(corr_frames.std(0))
    corr = torch.stack(parts, 1).reshape(1, -1, *corr_frames.shape[-2:])
    corr = F.relu(F.conv2d(corr, g("corr_encoder.0.weight"), g("corr_encoder.0.bias")))
    corr = F.relu(F.conv2d(corr, g("corr_encoder.2.weight"), g("corr_encoder.2.bias"), padding=1))
    x = torch.cat([inp, d, corr], 1)
    hx = torch.cat([net, x], 1)
    z = torch.sigmoid(F.conv2d(hx, g("gru.convz.weight"), g("gru.convz.bias"), padding=1))
    r = torch.sigmoid(F.conv2d(hx, g("gru.convr.weight"), g("gru.convr.bias"), padding=1))
    q = torch.tanh(F.conv2d(torch.cat([r * net, x], 1), g("gru.convq.weight"), g("gru.convq.bias"), padding=1))
    net = (1 - z) * net + z * q
    t = F.relu(F.conv2d(net, g(f"delta{stage}.0.weight"), g(f"delta{stage}.0.bias"), padding=1))
    delta = 0.01 * F.conv2d(t, g(f"delta{stage}.2.weight"), g(f"delta{stage}.2.bias"), padding=1)
    return net, delta


# ------------------------------------------------------------------------------ whole path


def resolve_cascade(cascade, num_levels=3, radius=5):
    """core/raft.py:76-81: (D,N,T) with D=-1 -> (2r+1)*2^(L-1); incre = 0.0025/N."""
    out = []
    for D, N, T in cascade:
        if D == -1:
            D = (2 * radius + 1) * 2 ** (num_levels - 1)
        out.append((D, 0.0025 / N, T))
    return out


def raft_forward(sd, images, poses, intrinsics, scale, cascade=((64, 64, 8), (-1, 320, 8)),
                 num_levels=3, radius=5, taps=None):
    """RAFT.forward in test mode on CPU fp32 (core/raft.py:34-108).  images [1,V+1,3,H,W] 0..255,
    poses [1,V+1,4,4], intrinsics [1,V+1,3,3].  Inputs are NOT mutated (the reference mutates, raft.py:35,40-41).
    ``taps``: optional dict that receives intermediate tensors for stage-wise parity tests."""
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    scale = float(torch.as_tensor(scale).reshape(-1)[0])
    poses = poses[0].clone().float()
    poses[:, :3, 3] *= scale
    intr = intrinsics[0].clone().float()
    intr[:, :2] /= 4
    imgs = images[0].float() * (2 / 255.0) - 1
    N, _, H, W = imgs.shape
    h, w = H // 4, W // 4
    ctx = encoder(imgs[0:1], sd, "cnet.", "none")
    net = torch.tanh(ctx[:, :64])
    inp = torch.relu(ctx[:, 64:])
    fmaps = torch.cat([encoder(imgs[i:i + 1], sd, "fnet.", "instance") for i in range(N)], 0)
    disp = torch.zeros(1, 1, h, w)
    if taps is not None:
        taps.update(net0=net, inp=inp, fmaps=fmaps)
    for stage, (D, incre, T) in enumerate(resolve_cascade(cascade, num_levels, radius)):
        vol, origin = cost_volume(fmaps, poses, intr, D, incre, disp[0, 0], shift=(stage == 0))
        levels = pyramid(vol, num_levels)
        if taps is not None:
            taps[f"vol{stage}"] = vol
            taps[f"origin{stage}"] = origin
        for it in range(T):
            feats = lookup(levels, origin, disp[0, 0], D, incre, radius)
            net, delta = update_block(sd, net, inp, disp, feats, stage)
            disp = disp + delta
            if taps is not None and it == 0:
                taps[f"feats{stage}"] = feats
                taps[f"net{stage}"] = net
                taps[f"delta{stage}"] = delta
    return disp * scale


# ------------------------------------------------------- caller-side transforms (row 14)


def scale_operation(images, intrinsics, s):
    """utils/data_utils.py:58-66 (out of place). images [N,3,H,W], intrinsics [N,3,3]."""
    ht, wd = int(s * images.shape[2]), int(s * images.shape[3])
    intrinsics = intrinsics.clone()
    intrinsics[:, 0] *= s
    intrinsics[:, 1] *= s
    return F.interpolate(images, [ht, wd], mode="bilinear", align_corners=True), intrinsics


def crop_operation(images, intrinsics, crop_h, crop_w):
    """utils/data_utils.py:69-78 (out of place)."""
    x0 = (images.shape[3] - crop_w) // 2
    y0 = (images.shape[2] - crop_h) // 2
    intrinsics = intrinsics.clone()
    intrinsics[:, 0, 2] -= x0
    intrinsics[:, 1, 2] -= y0
    return images[:, :, y0:y0 + crop_h, x0:x0 + crop_w], intrinsics


def disp_to_depth(disp):
    """inference.py:57-58: depth = 0 where disp == 0 else 1/disp (float32)."""
    return torch.where(disp == 0, torch.zeros_like(disp), 1.0 / disp)


def pfm_bytes(depth):
    """utils/frame_utils.py:138-163 for a 2-D little-endian float32 map: 'Pf', 'W H', '-1.000000', rows bottom-up."""
    import numpy as np
    a = np.ascontiguousarray(np.flipud(depth.detach().cpu().numpy().astype("<f4")))
    return b"Pf\n" + b"%d %d\n" % (a.shape[1], a.shape[0]) + b"%f\n" % -1.0 + a.tobytes()
