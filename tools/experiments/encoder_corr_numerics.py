#!/usr/bin/env python3
"""Numerics experiment (CPU, oracle side only; round 6): what does it cost end to end if the two CORRECTION terms of the encoders' split-f16
products  x*w ~ xh*wh + 2^-11 (xh*wl' + xl'*wh)  (csrc/enc_pc.hip: three f16 MFMAs per product) run on the 8- / 6-bit forms of
v_mfma_scale_f32_32x32x64_f8f6f4 - as the update block's do since rounds 3 / 6?
    "f16x3"   : as shipped - all three terms with f16 operands
    "f16"     : xh*wh only (no correction terms at all: the floor of what can go wrong)
    "fp8"     : corrections in e4m3, activations with a FIXED power-of-two scale (post-instance-norm values are O(1)), weights per tensor
    "fp6fix<s>": corrections in e2m3 (FP6), activations with the FIXED scale s (value / s is encoded; e2m3 holds 0, 0.125 .. 7.5, saturating),
                weights with one power-of-two scale per (output channel, tap, 16-channel block) from the block maximum (host side, free)
    "fp6blk"  : as above with a per-(pixel, 16-channel block) scale from the block maximum for the activations too (what conv_s16.hip does)
Only the convolutions enc_pc.hip runs are touched (3x3 and 1x1 convs of the trunk and the heads: Cin >= 32); the 7x7 stem keeps three f16 terms.
Reports relative L1 of the final disparity and of the feature maps against the fp32 oracle (bar of the project: 1e-4; auto gate: 2.5e-5).
usage: encoder_corr_numerics.py [H W V T [seed]]      (default 480 640 2 4 = cfg1's shape)"""
import os
import sys
import time
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import cer_oracle as O                                    # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene   # noqa: E402
from cer_mvs_amd import RAFT                                           # noqa: E402

MODE = "f32"
SAT = {}


def q16(t):
    return t.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def e2m3(a):
    """round-to-nearest-even onto e2m3's magnitudes (0, 0.125 .. 0.875, 1 .. 1.875 step 0.125, 2 .. 3.75 step 0.25, 4 .. 7.5 step 0.5), saturating"""
    s = torch.sign(a)
    a = a.abs()
    step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    return s * (torch.round(a / step) * step).clamp_max(7.5)


def chunk_perm(C):
    """channel order in which 16 consecutive entries form one K block of the instruction: {8kg .. 8kg+7} of k16-step 0 and of k16-step 1 of a 32-channel chunk"""
    idx = []
    for ch in range(C // 32):
        for kg in range(2):
            for ks in range(2):
                idx += [ch * 32 + ks * 16 + kg * 8 + e for e in range(8)]
    return torch.tensor(idx)


def q6_block(hi, lo, dim):
    """e2m3 of [hi | lo] with one power-of-two scale per block of 16 channels (permuted order) along `dim`, from the block's largest |hi| or |lo|"""
    C = hi.shape[dim]
    p = chunk_perm(C)
    inv = torch.argsort(p)
    h = hi.index_select(dim, p).movedim(dim, -1)
    l = lo.index_select(dim, p).movedim(dim, -1)
    hb, lb = h.reshape(*h.shape[:-1], -1, 16), l.reshape(*l.shape[:-1], -1, 16)
    m = torch.maximum(hb.abs().amax(-1, keepdim=True), lb.abs().amax(-1, keepdim=True)).clamp_min(1e-30)
    e = torch.floor(torch.log2(m)) - 2
    e = torch.where(m / torch.exp2(e) > 7.75, e + 1, e)
    s = torch.exp2(e)
    hq = (e2m3(hb / s) * s).reshape(h.shape).movedim(-1, dim).index_select(dim, inv)
    lq = (e2m3(lb / s) * s).reshape(l.shape).movedim(-1, dim).index_select(dim, inv)
    return hq, lq


def q8(t, scale):
    return (t * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) / scale


def conv_emul(x, w, b, **kw):
    """x: the normalised fp32 input of the conv (what the producers split), w, b: fp32 parameters"""
    if MODE == "f32" or w.shape[1] < 32:
        return F.conv2d(x, w, b, **kw)
    xh, wh = q16(x), q16(w)
    xl, wl = q16((x - xh) * 2048.0), q16((w - wh) * 2048.0)
    xd, wd = xh.double(), wh.double()
    main = F.conv2d(xd, wd, None, **kw)
    if MODE == "f16":
        corr = 0.0
    elif MODE == "f16x3":
        corr = F.conv2d(xd, wl.double(), None, **kw) + F.conv2d(xl.double(), wd, None, **kw)
    elif MODE == "fp8":
        sx = 16.0                                                    # activations: value * 16 -> e4m3 (normals from 2^-10, saturating at 28)
        m = float(torch.maximum(wh.abs().max(), wl.abs().max()))
        sw = 2.0 ** torch.floor(torch.log2(torch.tensor(448.0 / m))).item()
        corr = F.conv2d(q8(xh, sx).double(), q8(wl, sw).double(), None, **kw) + F.conv2d(q8(xl, sx).double(), q8(wh, sw).double(), None, **kw)
        SAT[MODE] = max(SAT.get(MODE, 0.0), float((xh.abs() > 28.0).float().mean()))
    elif MODE.startswith("fp6"):
        # weights [wl' | wh]: blocks of 16 input channels of one (output channel, tap): dim 1
        whq, wlq = q6_block(wh, wl, 1)
        if MODE == "fp6blk":
            xhq, xlq = q6_block(xh, xl, 1)
        else:
            s = float(MODE[6:])
            xhq, xlq = e2m3(xh / s) * s, e2m3(xl / s) * s
            SAT[MODE] = max(SAT.get(MODE, 0.0), float((xh.abs() > 7.75 * s).float().mean()))
        corr = F.conv2d(xhq.double(), wlq.double(), None, **kw) + F.conv2d(xlq.double(), whq.double(), None, **kw)
    else:
        raise ValueError(MODE)
    out = main + corr / 2048.0
    if b is not None:
        out = out + b.double().view(1, -1, 1, 1)
    return out.float()


def norm(y, kind):
    if kind != "instance":
        return y
    mean = y.mean(dim=(2, 3), keepdim=True)
    var = y.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (y - mean) * torch.rsqrt(var + 1e-5)


def res_block(x, sd, p, kind, stride):
    y = F.relu(norm(conv_emul(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1), kind))
    y = F.relu(norm(conv_emul(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), kind))
    if stride != 1:
        x = norm(conv_emul(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride), kind)
    return F.relu(x + y)


def encoder_q(x, sd, prefix, kind):
    x = F.relu(norm(F.conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], stride=2, padding=3), kind))
    x = res_block(x, sd, prefix + "layer1.0.", kind, 1)
    x = res_block(x, sd, prefix + "layer1.1.", kind, 1)
    x = res_block(x, sd, prefix + "layer2.0.", kind, 2)
    x = res_block(x, sd, prefix + "layer2.1.", kind, 1)
    return conv_emul(x, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])


def main():
    global MODE
    H, W, V, T = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (480, 640, 2, 4)))
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    modes = sys.argv[6].split(",") if len(sys.argv) > 6 else ["f16x3", "f16", "fp8", "fp6blk", "fp6fix1", "fp6fix0.5", "fp6fix2"]
    cascade = [(64, 64, T), (-1, 320, T)]
    model = RAFT(cascade=cascade, test_mode=True)
    sd = fill_state_dict(model.state_dict(), seed=seed)
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
    torch.set_num_threads(16)
    real = O.encoder
    outs, fm = {}, {}
    O.encoder = encoder_q
    try:
        for mode in ["f32"] + modes:
            MODE = mode
            t0 = time.time()
            taps = {}
            with torch.no_grad():
                outs[mode] = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade, taps=taps).double()
            fm[mode] = (taps["fmaps"].double(), taps["inp"].double(), taps["net0"].double())
            print(f"{mode:9s} done in {time.time() - t0:.1f} s", flush=True)
    finally:
        O.encoder = real
    ref = outs["f32"]
    rel = lambda a, b: float((a - b).abs().sum() / b.abs().sum())
    print(f"# {H}x{W}, {V} source views, {T}+{T} iterations, weights seed {seed}")
    for mode in modes:
        print(f"{mode:9s} disparity rel-L1 vs fp32 {rel(outs[mode], ref):.3e}  max {float((outs[mode] - ref).abs().max() / ref.abs().max()):.3e}"
              f"   | fmaps {rel(fm[mode][0], fm['f32'][0]):.3e}  inp {rel(fm[mode][1], fm['f32'][1]):.3e}  net0 {rel(fm[mode][2], fm['f32'][2]):.3e}"
              f"   saturated activations {SAT.get(mode, 0.0):.2e}")


if __name__ == "__main__":
    main()
