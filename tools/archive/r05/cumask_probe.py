#!/usr/bin/env python3
"""Which CUs does a HIP stream created with hipExtStreamCreateWithCUMask run on (MI355X, 8 XCDs x 32 CUs)?  Launches the cycle-stamp build of the
z|r conv (variants/libcermvs_sxtrace.so) on masked streams and counts blocks per (XCC, CU) from the recorded HW_ID / XCC_ID."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from cer_mvs_amd import _lib as L, ops

hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

dev = torch.device("cuda")
h, w = 296, 400
P = h * w
g = torch.Generator().manual_seed(0)
rnd = lambda *s, lo=-1.0, hi=1.0: (lo + (hi - lo) * torch.rand(*s, generator=g))
U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
net = ops.to_frag16(torch.tanh(rnd(P, 64, lo=-2, hi=2)).to(dev), h, w, U)
c2 = ops.to_frag16(torch.relu(rnd(P, 64, lo=-1, hi=2)).to(dev), h, w, R)
disp = rnd(P, lo=0.0005, hi=0.0025).to(dev)
pc = ops.PackedConvS16(rnd(128, 177, 3, 3, lo=-0.05, hi=0.05), None, [(64, 2, U), (49, 1, Dp), (64, 2, R)], dev, corr_fp8=True)
init = ops.s16_layout(rnd(P, 128, lo=-0.3, hi=0.3).to(dev), h, w, L.S16_ACC32)
PP = ops.s16_pixels(h, w)
nblk = ((h + 7) // 8) * ((w + 15) // 16)
trace = torch.zeros(PP, 64, device=dev)
z, rn = torch.rand(PP, 64, device=dev), torch.empty(PP, 64, device=dev)
run = lambda: ops.conv3x3_s16(pc, [net, disp, c2], h, w, L.EPI_GATES, out=z, out2=rn, aux=net, aux2=trace, init=init, log2s_out=U, log2s_aux=U)

def where(stream, label):
    with torch.cuda.stream(stream):
        run(); run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1)[:nblk * 4 * 32 * 2].view(np.uint64).reshape(nblk, 4, 32).astype(np.int64)
    hw, xcc = t[:, 0, 1], t[:, 0, 2] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4)      # CU id inside the XCC (SA/SE bits folded in roughly)
    se = (hw >> 13) & 7
    per_xcc = {int(x): len(set(zip(se[xcc == x].tolist(), cu[xcc == x].tolist()))) for x in np.unique(xcc)}
    print(f"{label}: {e0.elapsed_time(e1) * 1e3:.0f} us; distinct (SE, CU) per XCC: {per_xcc}; total {sum(per_xcc.values())}")

where(torch.cuda.current_stream(), "default stream")
full = [0xFFFFFFFF] * 8
where(masked_stream(full), "mask: 256 bits set")
where(masked_stream([0xFFFFFFFF] + [0] * 7), "mask: bits 0-31")
where(masked_stream([0x000000FF] * 8), "mask: low 8 bits of every word")
where(masked_stream([0x55555555] * 8), "mask: every second bit")
where(masked_stream([0xFFFFFFFF] * 3 + [0] * 5), "mask: words 0-2")
