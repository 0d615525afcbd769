#!/usr/bin/env python3
"""Whole forwards of the bench workload with EVERY lookup launch done twice on the same inputs - the compile-time specialisation and the
runtime-loop form of the experimental library (out_split | 0x100) - and compared on the device.  Answers: when a forward goes wrong
under GPU sharing, is the specialised lookup's output wrong for its inputs, or does the difference arise elsewhere?
usage (several concurrent copies): CER_MVS_LIB=cer-mvs_amd/csrc/variants/libcermvs_lkspec.so python tools/archive/repro_lookup_ab.py [forwards] [tag]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT, ops, update, _lib as L
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tag = sys.argv[2] if len(sys.argv) > 2 else str(os.getpid())
update.USE_PLANS = os.environ.get("AB_PLANS", "0") == "1"
dev = torch.device("cuda")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "e2e_cfg2.npz"))
H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
casc = [tuple(int(x) for x in c) for c in g["cascade"]]
images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
ref = torch.from_numpy(g["disp"]).to(dev).double()
model = RAFT(cascade=casc, test_mode=True, gru_precision="s16f8")
model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
model = model.to(dev).eval()
x = (images.to(dev), poses.to(dev), intr.to(dev))

orig = ops.lookup_encode
flags, dumped = [], [False]
NDUMP = int(os.environ.get('AB_NDUMP', '1'))
ndumped = [0]
AB = os.environ.get("AB_COMPARE", "1") == "1"
tmp = {}

ORDER = os.environ.get("AB_ORDER", "spec_first")       # spec_first | generic_first | spec_twice
SYNC = os.environ.get("AB_SYNC", "0") == "1"             # host synchronisation in front of the first launch of the pair

def patched(vol, origin, disp, w_t, b, D, incre, num_levels, radius, out=None, out_split=False, log2s=0, img_w=0):
    if SYNC:
        torch.cuda.synchronize()
    if AB and ORDER == "generic_first":
        t = tmp.get(out.shape)
        if t is None:
            t = tmp[out.shape] = torch.zeros_like(out)
        orig(vol, origin, disp, w_t, b, D, incre, num_levels, radius, out=t, out_split=int(out_split) | 0x100, log2s=log2s, img_w=img_w)
    o = orig(vol, origin, disp, w_t, b, D, incre, num_levels, radius, out=out, out_split=out_split, log2s=log2s, img_w=img_w)
    if AB:
        if ORDER != "generic_first":
            t = tmp.get(o.shape)
            if t is None:
                t = tmp[o.shape] = torch.zeros_like(o)
            orig(vol, origin, disp, w_t, b, D, incre, num_levels, radius, out=t, out_split=int(out_split) | (0 if ORDER == "spec_twice" else 0x100),
                 log2s=log2s, img_w=img_w)
        d = (o != t)
        flags.append(d.any())
        if not dumped[0]:
            cnt = d.sum()
            # (a dump needs a sync; only when something differs - checked lazily below)
            pending.append((cnt, d, o.clone(), t.clone(), vol, origin.clone(), disp.clone(), D, incre, int(out_split), log2s, img_w))
    return o

pending = []
prev_disp = [None]
ops.lookup_encode = patched
update.ops.lookup_encode = patched
errs, nbad_fw, nbad_lk = [], 0, 0
first = None
with torch.no_grad():
    for i in range(n):
        flags.clear(); pending.clear()
        o = model(*x, scale=scale)
        e = float((o.double() - ref).abs().sum() / ref.abs().sum())
        errs.append(e)
        if first is None:
            first = o.clone()
        same = bool(torch.equal(o, first))
        nb = int(torch.stack(flags).sum()) if flags else 0
        nbad_lk += nb
        if not same:
            nbad_fw += 1
        if (not same or nb) and nbad_fw + nbad_lk <= 12:
            print(f"[{tag}] forward {i}: equal to forward 0: {same}; rel-L1 vs capture {e:.3e}; lookup launches where the two forms differ: {nb} of {len(flags)}", flush=True)
        if nb and not dumped[0]:
            for k, (cnt, d, oo, tt, vol, origin, disp, D, incre, osp, log2s, img_w) in enumerate(pending):
                if int(cnt):
                    rows = d.any(1).nonzero().flatten()
                    if ndumped[0] == 0:
                        print(f"[{tag}]   first differing launch: index {k} in the forward, {int(cnt)} elements in {rows.numel()} rows; rows {rows[:16].tolist()}", flush=True)
                    os.makedirs("gpurun_out/lkdump", exist_ok=True)
                    # the pixels behind the differing rows of the frag16 buffer: row -> (m-tile, image row parity) -> 16 pixels
                    byte = rows * 256
                    mt, q = byte // 8192, (byte % 1024) // 256
                    mtx = (img_w + 15) // 16
                    y = (mt // mtx) * 2 + (q & 1)
                    x0 = (mt % mtx) * 16
                    pix = torch.unique((y * img_w + x0)[:, None] + torch.arange(16, device=rows.device)[None, :])
                    pix = pix[pix < vol.shape[0]]
                    # the whole output rows of those pixels' m-tiles (all groups, hi | lo)
                    mts = torch.unique(mt)
                    allrows = (mts[:, None] * 32 + torch.arange(32, device=rows.device)[None, :]).flatten()
                    torch.save({"rows": rows.cpu(), "spec": oo[rows].cpu(), "generic": tt[rows].cpu(), "out_split": osp, "k": k, "D": D, "incre": incre,
                                "log2s": log2s, "img_w": img_w, "pix": pix.cpu(), "vol": vol[pix].cpu(), "disp": disp[pix].cpu(), "origin": origin[pix].cpu(),
                                "allrows": allrows.cpu(), "spec_all": oo[allrows].cpu(), "generic_all": tt[allrows].cpu(),
                                "w": model.update_block.corr_encoder[0].weight.detach().cpu(), "b": model.update_block.corr_encoder[0].bias.detach().cpu(),
                                "disp_prev": (prev_disp[0][pix].cpu() if prev_disp[0] is not None else None)},
                               f"gpurun_out/lkdump/dump_{tag}.pt")
                    dumped[0] = True
                    break
print(f"[{tag}] {n} forwards: {nbad_fw} differ from forward 0; lookup A/B mismatching launches: {nbad_lk}; rel-L1 vs capture min {min(errs):.3e} max {max(errs):.3e}")
