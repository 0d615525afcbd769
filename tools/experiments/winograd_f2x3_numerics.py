#!/usr/bin/env python3
"""Costing of VERDICT r4 item 1(iv): Winograd F(2x2, 3x3) for the tensor-source groups of the update block's convolutions - the NUMERICS half
(CPU emulation; the register / LDS half is arithmetic on the kernel's own budget).

A 3x3 convolution of a [C = 64, H, W] activation tensor of the update block's value classes (hidden state |x| <= 1, ReLU features), weights of the
bench model's magnitude, computed (a) directly and (b) as F(2x2, 3x3) (input transform B^T d B in fp32, weights G g G^T transformed in fp64 on the
host, 16 element-wise channel contractions, output transform A^T m A in fp32), both with the library's split-f16 operand arithmetic: every fp32
operand x -> hi = f16(x s), lo = f16(x s - hi), products hi*hi + hi*lo + lo*hi exact, fp32 accumulation (the "s16" form; the fp8-correction form
adds ~2^-15 per product to either).  Error against an fp64 direct convolution, relative to sum |x||w| per output."""
import numpy as np

rng = np.random.default_rng(0)
C, K, H, W = 64, 64, 18, 34                      # one 16 x 32 output tile with its halo


def split(x, s):
    xs = (x * s).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def contract(xa, wa, sx, sw):
    """sum_c x[c, ...] * w[k, c, ...] in the split-f16 arithmetic; fp32 accumulator emulated by rounding the three-term sum per channel step."""
    xh, xl = split(xa, sx)
    wh, wl = split(wa, sw)
    acc = np.zeros((wa.shape[0],) + xa.shape[1:], dtype=np.float32)
    for c in range(xa.shape[0]):
        t = wh[:, c, None, None] * xh[c] + wh[:, c, None, None] * xl[c] + wl[:, c, None, None] * xh[c]
        acc = (acc.astype(np.float64) + t).astype(np.float32)
    return acc.astype(np.float64) / (sx * sw)


for name, gen, sx in (("hidden state (tanh-like, |x| <= 1)", lambda: np.tanh(rng.normal(0, 1.0, (C, H, W))), 2.0 ** 14),
                      ("ReLU features (0 .. ~3)", lambda: np.maximum(rng.normal(0.5, 1.0, (C, H, W)), 0), 2.0 ** 4)):
    x = gen().astype(np.float32).astype(np.float64)
    w = rng.uniform(-0.05, 0.05, (K, C, 3, 3)).astype(np.float32).astype(np.float64)
    Ho, Wo = H - 2, W - 2
    ref = np.zeros((K, Ho, Wo))
    mag = np.zeros((K, Ho, Wo))
    for dy in range(3):
        for dx in range(3):
            ref += np.einsum("kc,chw->khw", w[:, :, dy, dx], x[:, dy:dy + Ho, dx:dx + Wo])
            mag += np.einsum("kc,chw->khw", np.abs(w[:, :, dy, dx]), np.abs(x[:, dy:dy + Ho, dx:dx + Wo]))
    sw = 2.0 ** np.floor(np.log2(16384.0 / np.abs(w).max()))
    # (a) direct, split-f16
    direct = np.zeros((K, Ho, Wo))
    acc = np.zeros((K, Ho, Wo), dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            xh, xl = split(x[:, dy:dy + Ho, dx:dx + Wo], sx)
            wh, wl = split(w[:, :, dy, dx], sw)
            for c in range(C):
                t = wh[:, c, None, None] * xh[c] + wh[:, c, None, None] * xl[c] + wl[:, c, None, None] * xh[c]
                acc = (acc.astype(np.float64) + t).astype(np.float32)
    direct = acc.astype(np.float64) / (sx * sw)
    # (b) Winograd F(2x2, 3x3)
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    U = np.einsum("ia,kcab,jb->kcij", G, w, G)                       # fp64 on the host, then split like any weight
    swu = 2.0 ** np.floor(np.log2(16384.0 / np.abs(U).max()))
    ty, tx = Ho // 2, Wo // 2
    d = np.zeros((C, ty, tx, 4, 4), dtype=np.float32)
    for i in range(ty):
        for j in range(tx):
            d[:, i, j] = x[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4]
    V = np.einsum("ia,ctuab,jb->ctuij", Bt, d.astype(np.float64), Bt).astype(np.float32).astype(np.float64)     # input transform in fp32
    # transformed activations are up to 4 x larger: their split scale drops by 4 so that they stay inside the f16 range
    M = np.zeros((K, ty, tx, 4, 4))
    for i in range(4):
        for j in range(4):
            M[:, :, :, i, j] = contract(V[:, :, :, i, j], U[:, :, i, j], sx / 4.0, swu)
    Y = np.einsum("ia,ktuab,jb->ktuij", At, M.astype(np.float32).astype(np.float64), At)
    wino = Y.transpose(0, 1, 3, 2, 4).reshape(K, Ho, Wo)
    ed, ew = np.abs(direct - ref) / mag, np.abs(wino - ref) / mag
    print(f"{name}: error / sum|x||w|   direct split-f16: rms {np.sqrt((ed ** 2).mean()):.2e} max {ed.max():.2e}   "
          f"Winograd F(2x2,3x3) split-f16: rms {np.sqrt((ew ** 2).mean()):.2e} max {ew.max():.2e}   ratio {np.sqrt((ew ** 2).mean()) / np.sqrt((ed ** 2).mean()):.1f} x")
