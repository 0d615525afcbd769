import sys, os, warnings, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from cer_mvs_amd import RAFT
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
from conftest import rel_l1
dev = torch.device("cuda")
H, W, V = 160, 224, 3
cascade = [(64, 64, 16), (-1, 320, 16)]
def weights(gain, heavy):
    sd = fill_state_dict(RAFT(cascade=cascade, test_mode=True).state_dict(), seed=41)
    gen = torch.Generator().manual_seed(5)
    for k_, v in sd.items():
        if k_.startswith("update_block.") and k_.endswith("weight") and v.dim() == 4 and "delta" not in k_:
            v = v * gain
            if heavy: v = torch.where(torch.rand(v.shape, generator=gen) < 0.02, v * 8.0, v)
            sd[k_] = v
    return sd
def make(prec, sd):
    m = RAFT(cascade=cascade, test_mode=True, gru_precision=prec); m.load_state_dict(sd); m = m.to(dev).eval(); m.overflow_policy = "ignore"; return m
scenes = {}
for seed in (33, 7):
    im, po, it, sc = synthetic_scene(H, W, V, seed=seed); scenes[f"tex{seed}"] = (im, po, it, sc)
im, po, it, sc = scenes["tex33"]
scenes["flat"] = (torch.full_like(im, 127.0), po, it, sc)
scenes["lowcontrast"] = (127.0 + (im - 127.0) * 0.05, po, it, sc)
scenes["noise"] = (torch.rand(im.shape, generator=torch.Generator().manual_seed(1)) * 255, po, it, sc)
with torch.no_grad(), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for gain, heavy in ((1.0, False), (2.0, False), (1.0, True), (1.5, True), (2.0, True)):
        sd = weights(gain, heavy)
        m8, m16 = make("s16f8", sd), make("s16", sd)
        row = []
        for name, (a, b, c, s_) in scenes.items():
            x = (a.to(dev), b.to(dev), c.to(dev))
            row.append(f"{name} {rel_l1(m8(*x, scale=s_).cpu(), m16(*x, scale=s_).cpu()):.2e}")
        print(f"gain {gain} heavy {heavy}: " + "  ".join(row), flush=True)
