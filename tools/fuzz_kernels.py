#!/usr/bin/env python3
"""Randomised shape sweep of the HIP kernels against torch / the oracles (run on the GPU box; not part of the test suite):
python tools/fuzz_kernels.py [--iters N] [--seed S]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cer_mvs_amd import _lib as L, ops, fusion                            # noqa: E402
from cer_mvs_amd.corr import fmaps_to_nhwc                                 # noqa: E402
from oracle import cer_oracle as O                                        # noqa: E402
from oracle import fusion_oracle as FO                                    # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(args.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    rn = lambda *s: torch.randn(*s, generator=g)
    dev = torch.device("cuda")
    worst = {}

    def note(k, v, bar, ctx):
        worst[k] = max(worst.get(k, 0.0), v)
        if not (v < bar):
            print(f"FAIL {k}: {v:.3e} >= {bar:.1e}  {ctx}")
            raise SystemExit(1)

    for it in range(args.iters):
        # ---- 3x3 conv with a disparity source, all epilogues (interior tiles take the collapsed path when they exist)
        h, w = ri(3, 40), ri(3, 150)
        P = h * w
        net, c2 = torch.tanh(rn(1, 64, h, w)), torch.relu(rn(1, 64, h, w))
        disp = 0.0005 + 0.002 * torch.rand(1, 1, h, w, generator=g)
        feat = 100 * O.disp_features(disp)
        to_l = lambda t: t[0].permute(1, 2, 0).reshape(P, -1).contiguous().to(dev)
        for cout, epi in ((128, L.EPI_GATES), (64, L.EPI_GRU), (64, L.EPI_RELU)):
            wt = rn(cout, 177, 3, 3) * 0.04
            init = rn(1, cout, h, w) * 0.3
            ref = F.conv2d(torch.cat([net, feat, c2], 1).double(), wt.double(), None, padding=1) + init.double()
            pc = ops.PackedConv3x3(wt, None, [(64, 0), (49, 1), (64, 0)], dev)
            srcs = [to_l(net), disp.reshape(-1).to(dev), to_l(c2)]
            if epi == L.EPI_GATES:
                z, rh = ops.conv3x3(pc, srcs, h, w, epi, aux=to_l(net), init=to_l(init))
                sg = torch.sigmoid(ref)
                note("conv_gates_z", rel(z, sg[0, :64].permute(1, 2, 0).reshape(P, 64)), 5e-6, (h, w))
                note("conv_gates_rh", rel(rh, (sg[0, 64:] * net[0].double()).permute(1, 2, 0).reshape(P, 64)), 5e-6, (h, w))
            elif epi == L.EPI_GRU:
                zz = torch.sigmoid(rn(1, 64, h, w))
                out = ops.conv3x3(pc, srcs, h, w, epi, aux=to_l(net), aux2=to_l(zz), init=to_l(init))
                exp = (1 - zz.double()) * net.double() + zz.double() * torch.tanh(ref)
                note("conv_gru", rel(out, exp[0].permute(1, 2, 0).reshape(P, 64)), 5e-6, (h, w))
            else:
                out = ops.conv3x3(pc, srcs, h, w, epi, init=to_l(init))
                note("conv_relu", rel(out, torch.relu(ref)[0].permute(1, 2, 0).reshape(P, 64)), 5e-6, (h, w))
        # ---- the same convs on the default kernels (csrc/conv_s16.hip) in both arithmetic forms: all three terms in f16, and the
        # correction terms on the fp8 matrix instruction (gru_precision="s16f8"); tile height chosen at random
        U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
        fr = lambda t, k: ops.to_frag16(to_l(t), h, w, k)
        unf = lambda t, k: ops.from_frag16(t, h, w, k)
        for f8 in (False, True, 6):                    # (6: the FP6-correction form of round 6, gru_precision="s16f6")
            bar = 5e-5 if f8 else 5e-6
            tag = {False: "s16", True: "f8", 6: "f6"}[f8]
            ops.TILE_MT = [0, 2, 3, 4][ri(0, 3)]
            try:
                for cout, epi in ((128, L.EPI_GATES), (64, L.EPI_GRU), (64, L.EPI_RELU)):
                    wt = rn(cout, 177, 3, 3) * 0.04
                    init = rn(1, cout, h, w) * 0.3
                    ref = F.conv2d(torch.cat([net, feat, c2], 1).double(), wt.double(), None, padding=1) + init.double()
                    pc = ops.PackedConvS16(wt, None, [(64, 2, U), (49, 1, Dp), (64, 2, R)], dev, corr_fp8=f8)
                    srcs = [fr(net, U), disp.reshape(-1).to(dev), fr(c2, R)]
                    ini = ops.s16_layout(to_l(init), h, w, L.S16_ACC32)
                    if epi == L.EPI_GATES:
                        z, rh = ops.conv3x3_s16(pc, srcs, h, w, epi, aux=srcs[0], init=ini, log2s_out=U, log2s_aux=U)
                        sg = torch.sigmoid(ref)
                        note(f"{tag}_gates_z", rel(ops.s16_layout(z, h, w, L.S16_F32X8, inverse=True), sg[0, :64].permute(1, 2, 0).reshape(P, 64)), bar, (h, w))
                        note(f"{tag}_gates_rh", rel(unf(rh, U), (sg[0, 64:] * net[0].double()).permute(1, 2, 0).reshape(P, 64)), bar, (h, w))
                    elif epi == L.EPI_GRU:
                        zz = torch.sigmoid(rn(1, 64, h, w))
                        out = ops.conv3x3_s16(pc, srcs, h, w, epi, aux=srcs[0], aux2=ops.s16_layout(to_l(zz), h, w, L.S16_F32X8), init=ini,
                                              log2s_out=U, log2s_aux=U)
                        exp = (1 - zz.double()) * net.double() + zz.double() * torch.tanh(ref)
                        note(f"{tag}_gru", rel(unf(out, U), exp[0].permute(1, 2, 0).reshape(P, 64)), bar, (h, w))
                    else:
                        out = ops.conv3x3_s16(pc, srcs, h, w, epi, init=ini, log2s_out=R)
                        note(f"{tag}_relu", rel(unf(out, R), torch.relu(ref)[0].permute(1, 2, 0).reshape(P, 64)), bar, (h, w))
            finally:
                ops.TILE_MT = 0
        # ---- cost volume + fused pyramid + lookup against the C oracle path (torch oracle)
        h1, w1, V, D = ri(2, 14), ri(2, 22), ri(1, 4), [64, 44, 20, 8][ri(0, 3)]
        fm = rn(V + 1, 64, h1, w1)
        poses = torch.eye(4).repeat(V + 1, 1, 1)
        for v in range(1, V + 1):
            poses[v, 0, 3] = 20.0 * v * (1 if v % 2 else -1)
            poses[v, 1, 3] = 7.0 * (v - 2)
        intr = torch.tensor([[90.0, 0, w1 / 2.0], [0, 90.0, h1 / 2.0], [0, 0, 1]]).repeat(V + 1, 1, 1)
        d0 = 0.002 * torch.rand(h1, w1, generator=g)
        incre = 0.0025 / 64
        shift = bool(ri(0, 1))
        vol_o, org_o = O.cost_volume(fm, poses, intr, D, incre, d0, shift)
        from cer_mvs_amd.projective import pij_matrices
        Pij = pij_matrices(poses, intr, [0] * V, list(range(1, V + 1))).to(dev)
        f1 = fmaps_to_nhwc(fm[0:1].to(dev))[0]
        f2 = fmaps_to_nhwc(fm[1:].to(dev), border=2)
        vol, org = ops.cost_build(f1, f2, Pij, d0.reshape(-1).to(dev), D, incre, shift, h1, w1, 3, fold=True, pyramid_scale=1.0 / V)
        note("cost_build", rel(vol[:, :D], vol_o.mean(0)[:, :D] if vol_o.dim() == 3 else vol_o), 2e-5, (h1, w1, V, D))
        assert torch.equal(org.cpu(), org_o.reshape(-1)), "origin"
        # ---- fusion vote
        Hh, Ww, S = ri(8, 40), ri(8, 60), ri(1, 6)
        from cer_mvs_amd.synthetic import synthetic_depth_maps, synthetic_scene
        _, ps, ks, _ = synthetic_scene(Hh, Ww, S, seed=it)
        dm = synthetic_depth_maps(Hh, Ww, S, seed=it)
        srcs = list(range(1, S + 1))
        t1 = 4.0 * 10 ** float(torch.rand(1, generator=g) * 2 - 1)
        gm, est = fusion.vote(dm[0].to(dev), ks[0, 0], ps[0, 0], dm[srcs].to(dev), ks[0, srcs], ps[0, srcs], t1, t1 * 325.0)
        om, oe = FO.vote(dm[0], ks[0, 0], ps[0, 0], dm[srcs], ks[0, srcs], ps[0, srcs], t1, t1 * 325.0)
        mism = gm.bool().cpu() != om
        if mism.any():
            # a mismatch is legitimate only at a pixel sitting on a threshold (ulp-level difference of the fp32 evaluation order):
            # measure the smallest relative distance of that pixel's (dist, rel-depth) pair to any of the 9 thresholds, any view
            drep, xr, yr, _, _ = FO.reproject_with_depth(dm[0][None].repeat(S, 1, 1), ks[0, 0][None].repeat(S, 1, 1), ps[0, 0][None].repeat(S, 1, 1),
                                                          dm[srcs], ks[0, srcs], ps[0, srcs])
            yy, xx = torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij")
            dist = torch.sqrt((xr - xx[None]) ** 2 + (yr - yy[None]) ** 2)
            rl = torch.abs(drep - dm[0][None]) / dm[0][None]
            margin = torch.full((Hh, Ww), 1e9)
            for i in range(2, 11):
                m1 = ((dist - i / t1).abs() / (i / t1)).min(0).values
                m2 = ((rl - i / (t1 * 325.0)).abs() / (i / (t1 * 325.0))).min(0).values
                margin = torch.minimum(margin, torch.minimum(m1, m2))
            # (dist = |reprojected pixel - pixel| is a difference of O(W) coordinates: its fp32 error is ~1e-6 * W absolute, i.e.
            # up to ~1e-3 relative to a threshold of a few tenths of a pixel - that is the legitimate borderline band)
            note("fusion_mismatch_margin", float(margin[mism].max()), 3e-3, (Hh, Ww, S))
        note("fusion_mask_mismatch", float(mism.float().mean()), 2e-2, (Hh, Ww, S))
        note("fusion_depth", rel(est, oe), 1e-4, (Hh, Ww, S))
    # ---- whole forward at random small sizes against the torch oracle (encoders + both stages + GRU)
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    for it in range(max(1, args.iters // 10)):
        H, W, V = 4 * ri(9, 30), 4 * ri(9, 40), ri(1, 3)
        cascade = [(64, 64, ri(1, 2)), (-1, 320, ri(1, 2))]
        images, poses, intr, scale = synthetic_scene(H, W, V, seed=100 + it)
        model = RAFT(cascade=cascade, test_mode=True)
        sd = fill_state_dict(model.state_dict(), seed=50 + it)
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        with torch.no_grad():
            got = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).cpu()
            ref = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade)
        note("e2e_vs_oracle", rel(got, ref), 1e-4, (H, W, V, cascade))
    print("OK", {k: f"{v:.2e}" for k, v in worst.items()})


if __name__ == "__main__":
    main()
