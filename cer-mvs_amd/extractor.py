"""Feature / context encoders (reference: core/extractor.py:7-155, BasicEncoder type "HR"/"LR").

Per BASELINE.json's north_star the dense encoder convolutions stay on PyTorch-ROCm (MIOpen picks
MFMA kernels); this module only has to reproduce the reference's parameter names so that
``load_state_dict(strict=True)`` accepts reference checkpoints:
conv1, layer{1,2[,3]}.{0,1}.conv{1,2}, layer{2,3}.0.downsample.0, conv2.  Norm layers carry no
parameters (InstanceNorm2d defaults; "none" = identity); 'batch'/'group' are also accepted."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _make_norm(kind, planes):
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "group":
        return nn.GroupNorm(num_groups=planes // 8, num_channels=planes)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"unknown norm_fn {kind!r}")


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.kind = norm_fn
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.norm1 = _make_norm(norm_fn, planes)
        self.norm2 = _make_norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            self.norm3 = _make_norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        # (the fused kernels have no autograd and overwrite the conv output in place: inference only)
        if x.is_cuda and x.dtype == torch.float32 and self.kind in ("instance", "none") and not torch.is_grad_enabled():
            return self._forward_fused(x)
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)

    def _forward_fused(self, x):
        """Same arithmetic with the statistics / normalise / ReLU / add / ReLU passes fused into HIP kernels
        (csrc/encoder_ops.hip): 7 tensor passes per block instead of 15.  Convolutions stay on MIOpen."""
        from . import ops
        inorm = self.kind == "instance"
        y = self.conv1(x).contiguous()
        y = ops.norm_act(y, ops.plane_stats(y) if inorm else None, relu_a=True, out=y)
        y = self.conv2(y).contiguous()
        ys = ops.plane_stats(y) if inorm else None
        if self.downsample is not None:
            r = self.downsample[0](x).contiguous()
            rs = ops.plane_stats(r) if inorm else None
        else:
            r, rs = x.contiguous(), None
        return ops.norm_act(y, ys, res=r, res_stats=rs, relu_a=True, relu_out=True, out=y)


class BasicEncoder(nn.Module):
    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0, type="HR"):
        super().__init__()
        dim = 32
        self.norm_fn = norm_fn
        self.type = type
        self.conv1 = nn.Conv2d(3, dim, 7, stride=2, padding=3)
        self.norm1 = _make_norm(norm_fn, dim) if norm_fn != "group" else nn.GroupNorm(8, dim)
        self.layer1 = nn.Sequential(ResidualBlock(dim, dim, norm_fn, 1), ResidualBlock(dim, dim, norm_fn, 1))
        self.layer2 = nn.Sequential(ResidualBlock(dim, 2 * dim, norm_fn, 2), ResidualBlock(2 * dim, 2 * dim, norm_fn, 1))
        last = 2 * dim
        if type == "LR":
            self.layer3 = nn.Sequential(ResidualBlock(2 * dim, 4 * dim, norm_fn, 2), ResidualBlock(4 * dim, 4 * dim, norm_fn, 1))
            last = 4 * dim
        self.conv2 = nn.Conv2d(last, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():       # same init family as the reference (core/extractor.py:110-117)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        lead = x.shape[:-3]
        x = x.reshape((-1,) + tuple(x.shape[-3:]))
        if x.is_cuda and x.dtype == torch.float32 and self.norm_fn in ("instance", "none") and not torch.is_grad_enabled():
            from . import ops
            x = self.conv1(x).contiguous()
            x = ops.norm_act(x, ops.plane_stats(x) if self.norm_fn == "instance" else None, relu_a=True, out=x)
        else:
            x = F.relu(self.norm1(self.conv1(x)))
        x = self.layer2(self.layer1(x))
        if self.type == "LR":
            x = self.layer3(x)
        x = self.conv2(x)
        return x.reshape(tuple(lead) + tuple(x.shape[-3:]))
