#!/usr/bin/env python3
"""Minimal launcher for profiling single kernels at cfg2 shapes on random data (no encoders, no MIOpen):
usage: python tools/prof_conv.py [zr|q|d1|c2|lookup|stem|tail|build0|build1] [--reps N] [--mode f16x3|fp32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cer_mvs_amd import _lib as L, ops                                     # noqa: E402
from cer_mvs_amd.corr import fmaps_to_nhwc                                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="zr")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--mode", default="f16x3")
    ap.add_argument("--hw", default="296x400")
    ap.add_argument("--baseline-scale", type=float, default=1.0, help="build0/build1: scales the views' translations (0: every hypothesis hits the pixel's own texel)")
    args = ap.parse_args()
    h, w = (int(x) for x in args.hw.split("x"))
    P = h * w
    dev = torch.device("cuda")
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5)
    net, c2, inp = torch.tanh(r(P, 64)), torch.relu(r(P, 64)), torch.relu(r(P, 64))
    disp = (0.001 + 0.0005 * torch.rand(P, device=dev))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run(fn):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{args.what} [{args.mode}] {1e3 * e0.elapsed_time(e1) / args.reps:.1f} us")

    if args.what in ("zr", "q"):
        cout = 128 if args.what == "zr" else 64
        pc = ops.PackedConv3x3(torch.randn(cout, 177, 3, 3) * 0.03, None, [(64, 0), (49, 1), (64, 0)], dev)
        init = r(P, cout)
        if args.what == "zr":
            run(lambda: ops.conv3x3(pc, [net, disp, c2], h, w, L.EPI_GATES, aux=net, init=init, mode=args.mode))
        else:
            z = torch.sigmoid(r(P, 64))
            out = torch.empty(P, 64, device=dev)
            run(lambda: ops.conv3x3(pc, [net, disp, c2], h, w, L.EPI_GRU, out=out, aux=net, aux2=z, init=init, mode=args.mode))
    elif args.what in ("d1", "c2"):
        cout = 256 if args.what == "d1" else 64
        pc = ops.PackedConv3x3(torch.randn(cout, 64, 3, 3) * 0.05, torch.randn(cout) * 0.1, [(64, 0)], dev)
        out = torch.empty(P, cout, device=dev)
        run(lambda: ops.conv3x3(pc, [net], h, w, L.EPI_RELU, out=out, mode=args.mode))
    elif args.what == "d1f":
        pc = ops.PackedConv3x3(torch.randn(256, 64, 3, 3) * 0.05, torch.randn(256) * 0.1, [(64, 0)], dev)
        proj = ops.delta_proj_pack(torch.randn(1, 256, 3, 3) * 0.05, dev)
        T = torch.empty(2, 9, P, device=dev)
        run(lambda: ops.conv3x3(pc, [net], h, w, L.EPI_DELTA, out=T, aux=proj, mode="f16x3"))
    elif args.what == "dsum":
        T = r(2, 9, P)
        run(lambda: ops.delta_sum(T, 0.1, disp, h, w, want_delta=False))
    elif args.what == "tail":
        hid = torch.relu(r(P, 256))
        wt = r(9, 256)
        run(lambda: ops.delta_tail(hid, wt, 0.1, disp, h, w, want_delta=False))
    elif args.what == "lookup":
        # the shipped form (round 5): level-0-only rows (64 floats), frag16 output for the s16 convolutions, the previous iteration's
        # disparity update (18 tap-plane reads per pixel) riding on the launch; CER_PROF_LOOKUP_R04=1: round 4's form (112-float rows, no update)
        r04 = __import__("os").environ.get("CER_PROF_LOOKUP_R04") == "1"
        vol = r(P, 112 if r04 else 64)
        w0t, b0 = r(33, 64), r(64)
        out = torch.empty(ops.s16_pixels(h, w), 64, device=dev)
        org, dd = disp.clone(), disp + 0.0002 * torch.rand(P, device=dev)
        T = 0.001 * r(2, 9, P)
        run(lambda: ops.lookup_encode(vol, org, dd, w0t, b0, 64, 0.0025 / 64, 3, 5, out=out, out_split=2, log2s=L.S16_RELU, img_w=w,
                                      delta=None if r04 else (T, 0.0)))
    elif args.what == "stem":
        import ctypes
        lib = L.load()
        N, H, W = 11, 4 * h, 4 * w
        x = torch.rand(N, 3, H, W, device=dev) * 255.0
        wt = (torch.rand(32, 3, 7, 7) - 0.5) * 0.4
        packed = torch.empty(lib.cer_enc_stem_s16_packed_size(), dtype=torch.float16)
        k = ctypes.c_int(0)
        L.check(lib.cer_enc_stem_s16_pack(ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(packed.data_ptr()), ctypes.byref(k)), "pack")
        packed, b = packed.to(dev), r(32)
        ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty(N, ho * wo, 32, device=dev)
        part = torch.empty(N, lib.cer_enc_stem_s16_tiles(ho, wo), 32, 2, device=dev)
        run(lambda: L.check(lib.cer_enc_stem_s16(L.dev_ptr(x, "x"), L.dev_ptr(packed, "w", torch.float16), L.dev_ptr(b, "b"), L.dev_ptr(out, "out"),
                                                 L.dev_ptr(part, "part"), N, H, W, 1, int(k.value), L.cur_stream()), "stem"))
    elif args.what in ("build0", "build1"):
        V = 10
        f1 = r(P, 64) * 0.25
        f2 = fmaps_to_nhwc(torch.randn(V, 64, h, w, device=dev), border=2)
        Pij = torch.eye(4).repeat(V, 1, 1)
        for v in range(V):
            Pij[v, 0, 3] = args.baseline_scale * 30000.0 * (v + 1) * (1 if v % 2 else -1)
            Pij[v, 1, 3] = args.baseline_scale * 900.0 * (v - 4)
        Pij = Pij.to(dev)
        if args.what == "build0":
            run(lambda: ops.cost_build(f1, f2, Pij, torch.zeros(P, device=dev), 64, 0.0025 / 64, True, h, w, 3, fold=True))
        else:
            run(lambda: ops.cost_build(f1, f2, Pij, disp, 44, 0.0025 / 320, False, h, w, 3, fold=True))


if __name__ == "__main__":
    main()
