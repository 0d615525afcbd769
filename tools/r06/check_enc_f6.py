#!/usr/bin/env python3
"""GPU check of the encoders' FP6-correction form (csrc/enc_pc.hip, flags & 8): both nets at several sizes against the oracle and against the
three-term f16 form; end to end on the cfg1 fixture.  usage: python tools/r06/check_enc_f6.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT
from cer_mvs_amd.encoder_hip import HipEncoder
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
from oracle import cer_oracle as O

dev = torch.device("cuda")
rel = lambda a, b: float((a.double() - b.double()).abs().sum() / b.double().abs().sum())
for size in [(72, 104), (128, 160), (64, 96), (296, 400), (70, 132)]:
    for which in ("fnet", "cnet"):
        images, _, _, _ = synthetic_scene(size[0], size[1], 2, seed=8)
        model = RAFT(test_mode=True)
        sd = fill_state_dict(model.state_dict(), seed=13)
        model.load_state_dict(sd)
        x = images[0].float() * (2 / 255.0) - 1
        eng = HipEncoder(getattr(model, which), dev)
        with torch.no_grad():
            eng.f6 = False
            g3 = eng.forward_nchw(x.to(dev)).cpu()
            eng.f6 = True
            g6 = eng.forward_nchw(x.to(dev)).cpu()
            g6b = eng.forward_nchw(x.to(dev)).cpu()
            ref = O.encoder(x, sd, which + ".", "instance" if which == "fnet" else "none")
        print(f"{which} {size}: f16x3 vs oracle {rel(g3, ref):.2e}   f6 vs oracle {rel(g6, ref):.2e}   f6 vs f16x3 {rel(g6, g3):.2e}   "
              f"max {float((g6 - g3).abs().max() / g3.abs().max()):.2e}  repeat identical {bool((g6 == g6b).all())}  nan {bool(torch.isnan(g6).any())}", flush=True)

fx = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "e2e_cfg1.npz"))
print("fixture keys", list(fx.keys())[:12])
