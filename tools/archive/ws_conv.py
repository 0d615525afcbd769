#!/usr/bin/env python3
"""Wave-specialised conv (csrc/experimental/gru_ws.hip) vs the production f16x3 conv: equality and time at cfg2 shapes.
Build: make -C cer-mvs_amd/csrc variants/libcermvs_ws.so;  run: CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_ws.so python tools/archive/ws_conv.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import _lib as L, ops

dev = torch.device("cuda")
torch.manual_seed(0)
lib = L.load()
fn = lib.cer_conv3x3_ws_f16x3
fn.restype = ctypes.c_int

def ws(pc, srcs, h, w, epi, out):
    ci = L.ConvInputs(); ci.nsrc = len(srcs)
    for i, (t, (c, k)) in enumerate(zip(srcs, pc.sources)):
        ci.src[i] = t.data_ptr(); ci.ch[i] = c; ci.kind[i] = k
    rc = fn(ctypes.byref(ci), ctypes.c_void_p(pc.packed_x.data_ptr()), ctypes.c_void_p(pc.bias.data_ptr()) if pc.bias is not None else None,
            ctypes.c_void_p(out.data_ptr()), h, w, pc.cout, epi, L.cur_stream())
    assert rc == 0, rc
    return out

def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

for (h, w, cin, cout) in [(13, 70, 64, 64), (24, 100, 64, 128), (296, 400, 64, 64), (296, 400, 64, 256), (296, 400, 128, 128)]:
    P = h * w
    srcs = [torch.randn(P, 64, device=dev) * 0.5 for _ in range(cin // 64)]
    pc = ops.PackedConv3x3(torch.randn(cout, cin, 3, 3) * 0.05, torch.randn(cout) * 0.1, [(64, 0)] * (cin // 64), dev)
    ref = ops.conv3x3(pc, srcs, h, w, L.EPI_RELU, mode="f16x3")
    out = torch.full((P, cout), float("nan"), device=dev)
    ws(pc, srcs, h, w, L.EPI_RELU, out)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    same = torch.equal(out, ref)
    line = f"{h}x{w} {cin}->{cout}: max|diff| {err:.3e} equal={same}"
    if P > 10000:
        o2 = torch.empty(P, cout, device=dev)
        line += f"  old {timeit(lambda: ops.conv3x3(pc, srcs, h, w, L.EPI_RELU, mode='f16x3', out=o2)):.1f} us  ws {timeit(lambda: ws(pc, srcs, h, w, L.EPI_RELU, out)):.1f} us"
    print(line)
