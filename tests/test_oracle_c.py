"""The C restatement (oracle/cer_oracle.c: a loop-for-loop reading of correlation_kernel.cu:59-116 and of
CorrBlock) against the torch restatement (oracle/cer_oracle.py) and the reference captures.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import REPO, rel_l1
from oracle import cer_oracle as O
from test_oracle_golden import hashed


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(REPO, "oracle", "libceroracle.so"))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("r", [0, 1, 2])
def test_alt_corr_literal_vs_grid_sample(clib, r):
    B, N, H1, W1, H2, W2, C = 2, 3, 6, 9, 7, 5, 64
    f1, f2 = hashed((B, H1, W1, C), 1), hashed((B, H2, W2, C), 2)
    xy = torch.stack([hashed((B, N, H1, W1), 3, -2.5, W2 + 1.5), hashed((B, N, H1, W1), 4, -2.5, H2 + 1.5)], -1).contiguous()
    rd = 2 * r + 1
    out = torch.empty(B, N, rd * rd, H1, W1)
    clib.oracle_alt_corr_forward(_p(f1), _p(f2), _p(xy), _p(out), B, N, H1, W1, H2, W2, C, r)
    for kx in range(rd):
        for ky in range(rd):
            ref = O.alt_corr_forward(f1, f2, xy + torch.tensor([kx - r, ky - r], dtype=torch.float32))[:, :, 0]
            assert rel_l1(out[:, :, ky + rd * kx], ref) < 2e-6, (r, kx, ky)


def test_cost_volume_pyramid_lookup_vs_reference_capture(clib, golden):
    g = golden("corrblock")
    h1, w1, V = int(g["h1"]), int(g["w1"]), int(g["V"])
    P = h1 * w1
    fmaps = hashed((V + 1, 64, h1, w1), 11, -2.0, 2.0).contiguous()
    poses, intr = torch.from_numpy(g["poses"])[0], torch.from_numpy(g["intrinsics"])[0]
    Pij = O.pij_matrices(poses, intr, [0] * V, list(range(1, V + 1))).contiguous()
    for stage, (D, N, shift) in enumerate(((64, 64, True), (44, 320, False))):
        incre = 0.0025 / N
        disp_in = torch.from_numpy(g[f"disp_in{stage}"]).reshape(-1).contiguous()
        vol, origin = torch.empty(V, P, D), torch.empty(P)
        clib.oracle_cost_volume(_p(fmaps), _p(Pij), _p(disp_in), _p(vol), _p(origin), V, 64, h1, w1, D, ctypes.c_double(incre), int(shift))
        assert torch.equal(origin.view(h1, w1), torch.from_numpy(g[f"origin{stage}"]))
        levels = [vol]
        for lv in range(1, 3):
            n = levels[-1].shape[-1]
            nxt = torch.empty(V, P, n // 2)
            clib.oracle_pool(_p(levels[-1]), _p(nxt), ctypes.c_long(V * P), n)
            levels.append(nxt)
        for lv in range(3):
            assert rel_l1(levels[lv], torch.from_numpy(g[f"pyr{stage}_{lv}"])) < 2e-6, (stage, lv)
        zinv = torch.from_numpy(g[f"zinv{stage}"]).reshape(-1)
        c = torch.clamp_min((zinv - origin) / incre + D // 2, 0.0).repeat(V).contiguous()
        feats = []
        for lv in range(3):
            o = torch.empty(11, V * P)
            clib.oracle_lookup_level(_p(levels[lv].reshape(V * P, -1).contiguous()), _p(c), _p(o), ctypes.c_long(V * P), levels[lv].shape[-1], lv, 5)
            feats.append(o.view(11, V, P).permute(1, 0, 2))
        feats = torch.cat(feats, 1).reshape(1, V, 33, h1, w1)
        assert rel_l1(feats, torch.from_numpy(g[f"feats{stage}"])) < 2e-6
