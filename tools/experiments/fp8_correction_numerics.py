#!/usr/bin/env python3
"""Numerics experiment (CPU, oracle): what would the update block's 3x3 convolutions cost in accuracy if the two CORRECTION terms of the
split-f16 product  x*w ~ xh*wh + xh*wl + xl*wh  ran on the fp8 matrix pipe (2x the f16 rate on gfx950) instead of f16?
    mode "f16x3" : all three terms with f16 operands (what conv_s16.hip computes; fp32-class)
    mode "f16"   : xh*wh only (plain f16 operands: the reference's own autocast class)
    mode "fp8c"  : xh*wh in f16  +  fp8(xh)*fp8(wl) + fp8(xl)*fp8(wh)  (e4m3, per-tensor power-of-two scales)
Runs the oracle end to end with the update block's 3x3 convs replaced and reports the relative L1 of the disparity against fp32."""
import os, sys, time
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import cer_oracle as O                                    # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene   # noqa: E402
from cer_mvs_amd import RAFT                                           # noqa: E402

MODE = "fp32"
real_conv2d = F.conv2d


def pow2_scale(t, target):
    m = float(t.abs().max())
    if m == 0:
        return 1.0
    import math
    return 2.0 ** math.floor(math.log2(target / m))


def q16(t):
    return t.to(torch.float16).to(torch.float32)


def q8(t):
    s = pow2_scale(t, 448.0)
    return (t * s).to(torch.float8_e4m3fn).to(torch.float32) / s


ONLY = None      # restrict the emulation to convs with this weight shape[:2] (others: OTHERS)
OTHERS = "f16x3"


def q6(t, dim):
    """e2m3 (FP6: 1 sign, 2 exponent, 3 mantissa bits; values 0, 0.125 .. 0.875, 1 .. 1.875, 2 .. 3.75, 4 .. 7.5) with one power-of-two
    scale per block of 16 consecutive elements along `dim` (the K block of v_mfma_scale_f32_32x32x64_f8f6f4 is 32 = [hi | lo] halves of 16
    channels: the lo halves sit 2^-11 below and take the same scale after the fixed 2^11), round to nearest even, no saturation loss:
    the scale puts the block maximum into [4, 7.5] or, if it would land in (7.5, 8), into (3.75, 4]."""
    t = t.movedim(dim, -1)
    shp = t.shape
    pad = (-shp[-1]) % 16
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(*t.shape[:-1], -1, 16)
    m = b.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(m)) - 2
    e = torch.where(m / torch.exp2(e) > 7.5, e + 1, e)
    s = torch.exp2(e)
    a = (b / s).abs()
    step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    q = torch.round(a / step) * step          # torch.round: half to even
    q = q.clamp_max(7.5) * torch.sign(b) * s
    q = q.reshape(*t.shape)
    if pad:
        q = q[..., :shp[-1]]
    return q.movedim(-1, dim)


def conv_emul(x, w, b=None, **kw):
    if MODE == "fp32" or w.shape[-1] != 3 or w.shape[1] < 64:
        return real_conv2d(x, w, b, **kw)
    if ONLY is not None and tuple(w.shape[:2]) != ONLY:
        return conv_mode(x, w, b, OTHERS, **kw)
    return conv_mode(x, w, b, MODE, **kw)


def conv_mode(x, w, b, MODE, **kw):
    sx, sw = pow2_scale(x, 16384.0), pow2_scale(w, 16384.0)
    xs, ws = x * sx, w * sw
    xh, wh = q16(xs), q16(ws)
    xl, wl = q16(xs - xh), q16(ws - wh)
    d = torch.float64
    main = real_conv2d(xh.to(d), wh.to(d), None, **kw)
    if MODE == "f16":
        corr = 0
    elif MODE == "xh_w2":        # activations f16 (hi only), weights hi + lo: 2 MFMAs
        corr = real_conv2d(xh.to(d), wl.to(d), None, **kw)
    elif MODE == "x2_wh":        # activations hi + lo, weights f16 (hi only): 2 MFMAs
        corr = real_conv2d(xl.to(d), wh.to(d), None, **kw)
    elif MODE == "f16x3":
        corr = real_conv2d(xh.to(d), wl.to(d), None, **kw) + real_conv2d(xl.to(d), wh.to(d), None, **kw)
    elif MODE == "fp6j":
        # as the kernel does it (round 6): ONE scale per K block of the instruction = 16 channels x [hi | lo * 2^11], taken from the hi halves
        # (|lo * 2^11| <= |hi| element by element); activations per pixel, weights per output channel and tap
        def joint(hi, lo):
            both = torch.cat([hi.unsqueeze(2), (lo * 2048.0).unsqueeze(2)], 2)         # [N, C, 2, ...] -> blocks of 16 channels x 2
            n, c = both.shape[:2]
            pad = (-c) % 16
            if pad:
                both = torch.cat([both, torch.zeros(n, pad, *both.shape[2:])], 1)
            g = both.reshape(n, -1, 16, *both.shape[2:])                               # [N, G, 16, 2, ...]
            m = g.abs().amax(dim=(2, 3), keepdim=True).clamp_min(1e-30)
            e = torch.floor(torch.log2(m)) - 2
            e = torch.where(m / torch.exp2(e) > 7.75, e + 1, e)
            sc = torch.exp2(e)
            a = (g / sc).abs()
            step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
            q = (torch.round(a / step) * step).clamp_max(7.5) * torch.sign(g) * sc
            q = q.reshape(n, -1, *both.shape[2:])[:, :c]
            return q[:, :, 0], q[:, :, 1] / 2048.0
        xhq, xlq = joint(xh, xl)
        whq, wlq = joint(wh, wl)
        corr = real_conv2d(xhq.to(d), wlq.to(d), None, **kw) + real_conv2d(xlq.to(d), whq.to(d), None, **kw)
    elif MODE == "fp6c":
        # both correction terms in e2m3 with per-16-channel block scales (activations: per pixel; weights: per output channel and tap);
        # the lo halves are 2^11 up-scaled first (the kernel's fixed pre-scale) so that they share the hi halves' range
        corr = real_conv2d(q6(xh, 1).to(d), (q6(wl * 2048.0, 1) / 2048.0).to(d), None, **kw) + \
               real_conv2d((q6(xl * 2048.0, 1) / 2048.0).to(d), q6(wh, 1).to(d), None, **kw)
    else:
        corr = real_conv2d(q8(xh).to(d), q8(wl).to(d), None, **kw) + real_conv2d(q8(xl).to(d), q8(wh).to(d), None, **kw)
    out = ((main + corr) / (sx * sw)).to(torch.float32)
    return out if b is None else out + b.view(1, -1, 1, 1)


MODES = ("fp32", "f16x3", "fp8c", "fp6c", "fp6j")
PER_CONV_MODES = ()
PER_CONV_OTHERS = ("fp8c",)


def main():
    global MODE
    H, W, V = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (240, 320, 3)))
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    cascade = [(64, 64, T), (-1, 320, T)]
    model = RAFT(cascade=cascade, test_mode=True)
    sd = fill_state_dict(model.state_dict(), seed=5)
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
    torch.set_num_threads(8)
    outs = {}
    O.F.conv2d = conv_emul
    try:
        for mode in MODES:
            MODE = mode
            t0 = time.time()
            with torch.no_grad():
                outs[mode] = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade).double()
            print(f"{mode:6s} done in {time.time() - t0:.1f} s", flush=True)
    finally:
        O.F.conv2d = real_conv2d
    ref = outs["fp32"]
    global ONLY, OTHERS
    rel = lambda o: float((o - ref).abs().sum() / ref.abs().sum())
    for mode in MODES[1:]:
        print(f"{mode:6s} rel-L1 vs fp32: {rel(outs[mode]):.3e}", flush=True)
    # one conv class at a time in a cheaper form, the others as the shipped default (fp8 corrections)
    O.F.conv2d = conv_emul
    try:
        for others in PER_CONV_OTHERS:
            for mode in PER_CONV_MODES:
                for shape in ((64, 64), (64, 241), (256, 64), (1, 256)):
                    ONLY, MODE, OTHERS = shape, mode, others
                    with torch.no_grad():
                        o = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade).double()
                    print(f"{mode} only in convs {shape}, others {others}: rel-L1 vs fp32 {rel(o):.3e}", flush=True)
    finally:
        O.F.conv2d = real_conv2d
        ONLY, OTHERS = None, "f16x3"


if __name__ == "__main__":
    main()
