"""CorrBlock: epipolar cost volume + correlation pyramid + multi-level lookup
(reference: core/corr.py:45-143; DirectCorr/direct_corr core/corr.py:12-43).

Same constructor and ``__call__`` as the reference.  Construction launches ONE fused HIP kernel per
stage (projection + bilinear gather + 64-channel dot for all views and hypotheses) plus the pyramid
kernel, instead of V x (einsum, divide, permute, clamp, 2 copies, 1 CUDA kernel, permute) and 2
avg_pools; ``__call__`` launches one lookup kernel instead of 528 grid_samples + 36 cats.

``fold_views=True`` (extension, used by RAFT.forward's fast path) stores the view-MEAN volume
[P, row] instead of [V, P, row]: exact for aggregation == ["mean"] because lookup and pooling are
linear and every view is sampled at the same index."""
import torch

from . import ops
from .projective import pij_matrices


class DirectCorr(torch.autograd.Function):
    """autograd wrapper of alt_cuda_corr with radius 0 (reference: core/corr.py:12-25)."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords):
        ctx.save_for_backward(fmap1, fmap2, coords)
        return ops.alt_corr_forward(fmap1, fmap2, coords, 0)

    @staticmethod
    def backward(ctx, grad_output):
        fmap1, fmap2, coords = ctx.saved_tensors
        return ops.alt_corr_backward(fmap1, fmap2, coords, grad_output.contiguous(), 0)


def fmaps_to_nhwc(fmaps, border=0):
    """[N,C,h,w] -> [N,(h+2b)*(w+2b),C] * (1/8)  (core/corr.py:29-35: permute, /8.0, contiguous, float);
    ``border`` zero texels on every side (the cost-build kernel wants 2 on the source maps).  One HIP pass."""
    from . import _lib as L
    N, C, h, w = fmaps.shape
    x = fmaps.float().contiguous()
    out = torch.empty(N, (h + 2 * border) * (w + 2 * border), C, device=x.device, dtype=torch.float32)
    L.check(L.load().cer_nchw_to_nhwc_border_f32(L.dev_ptr(x, "fmaps"), L.dev_ptr(out, "out"), N, C, h, w, border, 0.125, L.cur_stream()),
            "nchw_to_nhwc_border")
    return out


class CorrBlock:
    def __init__(self, fmaps, poses, intrinsics, ii, jj, nIncre, incre, disps_input, shift, num_levels, radius,
                 test_mode=True, do_report=False, fold_views=False, view_weight=None):
        if not fmaps.is_cuda:
            raise RuntimeError("fmaps must be a CUDA tensor")
        batch, num_frames, ch, h1, w1 = fmaps.shape
        if batch != 1:
            raise RuntimeError("CorrBlock: batch must be 1 (the reference's lookup is only layout-correct for B=1, core/corr.py:107)")
        self.num_levels, self.radius, self.test_mode = num_levels, radius, test_mode
        self.nIncre, self.incre = nIncre, incre
        self.h1, self.w1 = h1, w1
        self.fold_views = fold_views
        ii_l = [int(x) for x in torch.as_tensor(ii).reshape(-1).tolist()]
        jj_l = [int(x) for x in torch.as_tensor(jj).reshape(-1).tolist()]
        if len(set(ii_l)) != 1:
            raise RuntimeError("CorrBlock: all pairs must share one reference view (ii constant), as in core/raft.py:45")
        self.num_views = len(jj_l)
        f1 = fmaps_to_nhwc(fmaps[0, ii_l[0]:ii_l[0] + 1])[0]
        f2 = fmaps_to_nhwc(fmaps[0, jj_l], border=2)
        Pij = pij_matrices(poses[0], intrinsics[0], ii_l, jj_l).to(fmaps.device)
        disp_in = disps_input.reshape(-1).float().contiguous()
        total_views = self.num_views if view_weight is None else view_weight
        fuse = fold_views and nIncre <= 64          # view-mean fold: scale + pooled levels written by the build's epilogue
        vol, origin = ops.cost_build(f1, f2, Pij, disp_in, nIncre, incre, shift, h1, w1, num_levels, fold=fold_views,
                                     pyramid_scale=(1.0 / total_views) if fuse else None)
        if not fuse:
            ops.pyramid(vol, nIncre, num_levels, scale=(1.0 / total_views) if fold_views else 1.0)
        self.volume = vol                                           # [V,P,rs] or [P,rs]
        self.origin = origin                                        # [P]
        self.disps_origin = origin.view(1, 1, 1, h1, w1)
        offs, lens, _ = ops.row_layout(nIncre, num_levels)
        # reference attribute: list of [B*V*P, 1, 1, W_i] (views into the packed rows)
        self.corr_pyramid = [vol[..., o:o + n].reshape(-1, 1, 1, n) for o, n in zip(offs, lens)]
        if do_report and not shift:
            report()

    def __call__(self, zinv):
        """zinv [B,V,h1,w1] -> [B,V,L*(2r+1),h1,w1] float32 contiguous (core/corr.py:102-143)."""
        batch, num, h1, w1 = zinv.shape
        if batch != 1 or h1 != self.h1 or w1 != self.w1:
            raise RuntimeError("CorrBlock.__call__: zinv shape does not match the volume")
        z = zinv.reshape(num, h1 * w1).float().contiguous()
        if self.fold_views:
            out = ops.corr_lookup(self.volume, self.origin, z[0].contiguous(), self.nIncre, self.incre, self.num_levels, self.radius)
            return out.view(1, 1, -1, h1, w1)
        if num != self.num_views:
            raise RuntimeError("CorrBlock.__call__: zinv must have one plane per source view")
        out = ops.corr_lookup(self.volume, self.origin, z, self.nIncre, self.incre, self.num_levels, self.radius, per_view_disp=True)
        return out.view(1, num, -1, h1, w1)


def report():
    """Peak-memory print (reference: utils/memory.py:5-11 shells out to nvidia-smi)."""
    mem = torch.cuda.max_memory_allocated() // (1024 * 1024)
    print(f"inference memory: {mem} MB")
