"""Training-mode forward and loss (reference: core/raft.py:34-109 with test_mode=False, core/corr.py:12-43,
67-78,102-143, core/update.py:87-120, loss.py:5-41).

The inference path of this package (raft.py) is a chain of fused, autograd-free HIP kernels.  Training needs gradients, so the
training-mode forward follows the reference's own control flow in differentiable torch ops - encoders (MIOpen convs), projection,
pyramid pooling, lookup (grid_sample), update block - around the ONE native op the reference has: the epipolar correlation,
``alt_cuda_corr`` under ``DirectCorr`` (corr.py), whose forward AND backward run on the HIP kernels of csrc/alt_corr.hip
(the backward deterministically: sorted segmented reduction, no float atomics).  Datasets and the optimiser loop are out of scope."""
import torch
import torch.nn.functional as F

from .corr import DirectCorr
from .projective import pij_matrices


def _coords(Pij, disps, h, w):
    """Source-view pixel coordinates of every (hypothesis, pixel) (utils/projective_ops.py:5-28; core/corr.py:86-88):
    Pij [V,4,4], disps [D,h,w] -> [V,D,h,w,2], clamped to +-1e4."""
    dev = disps.device
    y, x = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    x0 = torch.stack([x.expand_as(disps), y.expand_as(disps), torch.ones_like(disps), disps], -1)          # [D,h,w,4]
    x1 = torch.einsum("vkh,dyxh->vdyxk", Pij, x0)
    x1 = x1 / x1[..., 2:3]
    return x1[..., :2].clamp(min=-1e4, max=1e4).contiguous()


class TrainCorrBlock:
    """CorrBlock with test_mode=False (core/corr.py:46-99,102-143): level 0 through DirectCorr (differentiable w.r.t. the
    feature maps, like the reference: coords get no gradient), pooled levels and the lookup in torch autograd."""

    def __init__(self, fmaps, Pij, nIncre, incre, disps_input, shift, num_levels, radius):
        _, nv1, _, h1, w1 = fmaps.shape
        V = nv1 - 1
        dev = fmaps.device
        self.nIncre, self.incre, self.num_levels, self.radius, self.V, self.h1, self.w1 = nIncre, incre, num_levels, radius, V, h1, w1
        disps = ((torch.arange(nIncre) - nIncre // 2) * incre).to(dev).view(nIncre, 1, 1)
        d_in = disps_input.reshape(1, h1, w1).float()
        lim = torch.tensor(nIncre // 2 * incre, device=dev, dtype=torch.float32)
        self.origin = torch.where(d_in < lim, lim, d_in) if shift else d_in.clone()              # core/corr.py:59-62
        coords = _coords(Pij, disps + self.origin, h1, w1)                                       # [V,D,h,w,2]
        f = fmaps[0].permute(0, 2, 3, 1) / 8.0                                                   # core/corr.py:29-31
        f1 = f[:1].expand(V, -1, -1, -1).contiguous().float()
        f2 = f[1:].contiguous().float()
        corr = DirectCorr.apply(f1, f2, coords)                                                  # [V,D,1,h,w]
        corr = corr.permute(0, 2, 3, 4, 1).reshape(V * h1 * w1, 1, 1, nIncre)                   # core/corr.py:41-43
        self.corr_pyramid = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, [1, 2], stride=[1, 2])
            self.corr_pyramid.append(corr)

    def __call__(self, disp):
        """disp [1,1,h,w] -> [1,V,L*(2r+1),h,w] (core/corr.py:102-143, the train branch: all taps of a level in one sample)."""
        r, V, h1, w1 = self.radius, self.V, self.h1, self.w1
        zinv = disp.reshape(1, h1, w1, 1).expand(V, -1, -1, -1)
        coords = torch.clamp_min((zinv - self.origin.view(1, h1, w1, 1)) / self.incre + self.nIncre // 2, 0.0)
        out = []
        dx = torch.linspace(-r, r, 2 * r + 1, device=disp.device).view(1, 1, 2 * r + 1, 1)
        for i, corr in enumerate(self.corr_pyramid):
            x0 = dx + coords.reshape(V * h1 * w1, 1, 1, 1) / 2 ** i
            W = corr.shape[-1]
            grid = torch.cat([2 * x0 / (W - 1) - 1, torch.zeros_like(x0)], -1)                   # utils/bilinear_sampler.py:6-25
            s = F.grid_sample(corr, grid, align_corners=True)
            out.append(s.view(V, h1, w1, -1))
        return torch.cat(out, -1).permute(0, 3, 1, 2).reshape(1, V, -1, h1, w1).contiguous()


def update_block_torch(ub, net, inp, disp, corr_frames, stage):
    """UpdateBlock.forward in differentiable torch ops on the module's own parameters (core/update.py:87-120)."""
    cn, gn, dn = ub._names(stage)
    ce, gru, de = getattr(ub, cn), getattr(ub, gn), getattr(ub, dn)
    d = 100 * ub.disp_encoder(disp)
    parts = []
    if "mean" in ub.aggregation:
        parts.append(torch.mean(corr_frames, dim=1))
    if "max" in ub.aggregation:
        parts.append(torch.max(corr_frames, dim=1).values)
    if "std" in ub.aggregation:
        parts.append(torch.std(corr_frames, dim=1))
    corr = torch.stack(parts, dim=2).view(1, -1, *net.shape[-2:])
    corr = ce(corr)
    x = torch.cat([inp, d, corr], dim=1)
    hx = torch.cat([net, x], dim=1)
    z = torch.sigmoid(gru.convz(hx))
    r = torch.sigmoid(gru.convr(hx))
    q = torch.tanh(gru.convq(torch.cat([r * net, x], dim=1)))
    net = (1 - z) * net + z * q
    delta = 0.01 * de(net)
    return net, delta


def forward_train(model, images, poses, intrinsics, scale=None):
    """RAFT.forward with test_mode=False: returns the list of disparity predictions, one per GRU iteration, each [1,1,h,w]
    and NOT multiplied by ``scale`` (core/raft.py:103,109).  Inputs are not mutated.  fp32 (no autocast)."""
    if not images.is_cuda:
        raise RuntimeError("forward_train: images must be a CUDA tensor")
    batch, num, _, ht, wd = images.shape
    if batch != 1:
        raise RuntimeError("forward_train: batch must be 1")
    dev = images.device
    poses = poses.clone().float()
    if scale is not None:
        poses[..., :3, 3] *= float(torch.as_tensor(scale).reshape(-1)[0])
    factor = 8 if model.encoder_type == "LR" else 4
    intr = intrinsics.clone().float()
    intr[:, :, :2] /= factor
    imgs = images.float() * (2 / 255.0) - 1
    h, w = ht // factor, wd // factor
    V = num - 1
    ub = model.update_block
    with torch.enable_grad():
        ctx = model.cnet(imgs[:, [0]]).float()
        net, inp = ctx[0].split([model.dim_net, model.dim_inp], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        fmaps = model.fnet(imgs).float()
        Pij = pij_matrices(poses[0], intr[0], [0] * V, list(range(1, V + 1))).to(dev)
        disp = torch.zeros(1, 1, h, w, device=dev)
        predictions = []
        for stage, (D, incre, T) in enumerate(model.stages()):
            corr_fn = TrainCorrBlock(fmaps, Pij, D, incre, disp.detach(), stage == 0, ub.num_levels, ub.radius)
            for _ in range(T):
                disp = disp.detach()
                corr_frames = corr_fn(disp)
                net, delta = update_block_torch(ub, net, inp, disp, corr_frames, stage)
                disp = disp + delta.float()
                predictions.append(disp)
    return predictions


def sequence_loss(disp_est, disp_gt, depthloss_threshold=100, gradual_weight=None, gamma=0.9, depth_cut=1e-3):
    """Loss over the sequence of predictions (loss.py:5-41): exponentially weighted L1 on disparity blended with a clamped L1 on
    depth; returns (loss, metrics).  ``gradual_weight`` in [0,1] is required, as in the reference (gin supplies it there)."""
    if gradual_weight is None:
        raise ValueError("sequence_loss: gradual_weight is required (the reference binds it through gin)")
    n = len(disp_est)
    valid = disp_gt > 0.0
    ht, wd = disp_gt.shape[-2:]
    est = [F.interpolate(d, [ht, wd], mode="bilinear", align_corners=True) for d in disp_est]
    loss = 0.0
    for i in range(n):
        wgt = gamma ** (n - i - 1)
        loss_disp = (est[i] - disp_gt).abs()
        loss_depth = (1.0 / est[i].clamp(min=depth_cut) - 1.0 / disp_gt.clamp(min=depth_cut)).abs()
        loss_depth = loss_depth.clamp(max=depthloss_threshold) / 3.6e5
        i_loss = gradual_weight * loss_depth + (1 - gradual_weight) * loss_disp
        loss = loss + wgt * (valid * i_loss).mean() + 0.01 * wgt * i_loss.mean()
    epe = (1.0 / est[-1].clamp(min=depth_cut) - 1.0 / disp_gt).abs().view(-1)[valid.view(-1)]
    metrics = {"mean_depth_error": epe.mean().item(), "less3": (epe < 3).float().mean().item(),
               "less10": (epe < 10).float().mean().item(), "less25": (epe < 25).float().mean().item()}
    return loss, metrics
