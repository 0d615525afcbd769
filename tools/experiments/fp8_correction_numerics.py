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


ONLY = None      # restrict the emulation to convs with this weight shape[:2] (others: f16x3)


def conv_emul(x, w, b=None, **kw):
    if MODE == "fp32" or w.shape[-1] != 3 or w.shape[1] < 64:
        return real_conv2d(x, w, b, **kw)
    if ONLY is not None and tuple(w.shape[:2]) != ONLY:
        return conv_mode(x, w, b, "f16x3", **kw)
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
    else:
        corr = real_conv2d(q8(xh).to(d), q8(wl).to(d), None, **kw) + real_conv2d(q8(xl).to(d), q8(wh).to(d), None, **kw)
    out = ((main + corr) / (sx * sw)).to(torch.float32)
    return out if b is None else out + b.view(1, -1, 1, 1)


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
        for mode in ("fp32", "f16x3", "fp8c", "xh_w2", "x2_wh", "f16"):
            MODE = mode
            t0 = time.time()
            with torch.no_grad():
                outs[mode] = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade).double()
            print(f"{mode:6s} done in {time.time() - t0:.1f} s", flush=True)
    finally:
        O.F.conv2d = real_conv2d
    ref = outs["fp32"]
    global ONLY
    O.F.conv2d = conv_emul
    try:
        for shape in ((64, 64), (64, 241), (256, 64), (1, 256)):
            ONLY, MODE = shape, "xh_w2"
            with torch.no_grad():
                o = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade).double()
            print(f"xh_w2 only in convs {shape}: rel-L1 vs fp32 {float((o - ref).abs().sum() / ref.abs().sum()):.3e}", flush=True)
    finally:
        O.F.conv2d = real_conv2d
        ONLY = None
    for mode in ("f16x3", "fp8c", "xh_w2", "x2_wh", "f16"):
        print(f"{mode:6s} rel-L1 vs fp32: {float((outs[mode] - ref).abs().sum() / ref.abs().sum()):.3e}")


if __name__ == "__main__":
    main()
