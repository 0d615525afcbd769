#!/usr/bin/env python3
"""Numerics experiment (CPU, oracle side only): what does it cost end to end if the encoders' INTER-LAYER tensors
(raw conv outputs and the materialised residual merges) are stored narrower than fp32?
    "f32"    : as shipped (4 B per value: f16 hi | lo of the split operand)
    "f16"    : raw conv outputs / merges rounded to f16 (2 B), instance-norm statistics from the fp32 accumulators, products exact
    "f16in"  : as "f16", and the normalised conv INPUT rounded to f16 as well (two-term product xh*wh + xh*wl: no lo plane at all)
    "f16e4"  : f16 hi + e4m3 of the residual (3 B)
    "bf16x2" : for scale: hi + lo in bf16 (what 4 B buy today is ~22 bits; this is ~16)
Reports relative L1 of the final disparity against the fp32 oracle (bar of the project: 1e-4; gate for adoption: 2e-5).
usage: encoder_f16_numerics.py [H W V T]      (default 480 640 2 4 = cfg1's shape)"""
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


def q_store(t):
    """what a stored inter-layer tensor keeps"""
    if MODE == "f32":
        return t
    if MODE in ("f16", "f16in"):
        return t.to(torch.float16).to(torch.float32)
    if MODE == "f16e4":
        hi = t.to(torch.float16).to(torch.float32)
        r = t - hi
        # per-tensor power-of-two scale for the residual (|r| <= 2^-11 |t|)
        m = float(r.abs().max())
        s = 1.0 if m == 0 else 2.0 ** torch.floor(torch.log2(torch.tensor(448.0 / m))).item()
        return hi + (r * s).to(torch.float8_e4m3fn).to(torch.float32) / s
    if MODE == "bf16x2":
        hi = t.to(torch.bfloat16).to(torch.float32)
        return hi + (t - hi).to(torch.bfloat16).to(torch.float32)
    raise ValueError(MODE)


def q_in(t):
    return t.to(torch.float16).to(torch.float32) if MODE == "f16in" else t


def norm_q(y, kind):
    """statistics from the fp32 accumulators, applied to the stored (rounded) values"""
    yq = q_store(y)
    if kind != "instance":
        return yq
    mean = y.mean(dim=(2, 3), keepdim=True)
    var = y.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (yq - mean) * torch.rsqrt(var + 1e-5)


def res_block(x, sd, p, kind, stride):
    y = F.relu(norm_q(F.conv2d(q_in(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1), kind))
    y = F.relu(norm_q(F.conv2d(q_in(y), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), kind))
    if stride != 1:
        x = norm_q(F.conv2d(q_in(x), sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride), kind)
    return q_store(F.relu(x + y))        # (pessimistic: every merge is treated as materialised)


def encoder_q(x, sd, prefix, kind):
    x = F.relu(norm_q(F.conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], stride=2, padding=3), kind))
    x = res_block(x, sd, prefix + "layer1.0.", kind, 1)
    x = res_block(x, sd, prefix + "layer1.1.", kind, 1)
    x = res_block(x, sd, prefix + "layer2.0.", kind, 2)
    x = res_block(x, sd, prefix + "layer2.1.", kind, 1)
    return F.conv2d(q_in(x), sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])       # (the head's output keeps 4 B: split planes)


def main():
    global MODE
    H, W, V, T = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (480, 640, 2, 4)))
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    cascade = [(64, 64, T), (-1, 320, T)]
    model = RAFT(cascade=cascade, test_mode=True)
    sd = fill_state_dict(model.state_dict(), seed=seed)
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
    torch.set_num_threads(8)
    real = O.encoder
    outs, fm = {}, {}
    O.encoder = encoder_q
    try:
        for mode in ("f32", "f16", "f16in", "f16e4", "bf16x2"):
            MODE = mode
            t0 = time.time()
            taps = {}
            with torch.no_grad():
                outs[mode] = O.raft_forward(sd, images, poses, intr, scale, cascade=cascade, taps=taps).double()
            fm[mode] = (taps["fmaps"].double(), taps["inp"].double(), taps["net0"].double())
            print(f"{mode:7s} done in {time.time() - t0:.1f} s", flush=True)
    finally:
        O.encoder = real
    ref = outs["f32"]
    rel = lambda a, b: float((a - b).abs().sum() / b.abs().sum())
    for mode in ("f16", "f16in", "f16e4", "bf16x2"):
        print(f"{mode:7s} disparity rel-L1 vs fp32 {rel(outs[mode], ref):.3e}  max {float((outs[mode] - ref).abs().max() / ref.abs().max()):.3e}"
              f"   | fmaps {rel(fm[mode][0], fm['f32'][0]):.3e}  inp {rel(fm[mode][1], fm['f32'][1]):.3e}  net0 {rel(fm[mode][2], fm['f32'][2]):.3e}")


if __name__ == "__main__":
    main()
