import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_oracle_golden import hashed
from cer_mvs_amd import ops
from oracle import cer_oracle as O
dev = torch.device('cuda')
h1, w1, V, C, D, incre = 7, 13, 2, 64, 20, 0.0025 / 64
fm = hashed((V + 1, C, h1, w1), 81, -2, 2)
poses = torch.eye(4).repeat(V + 1, 1, 1); poses[1, 0, 3] = 40.0; poses[2, 0, 3] = 1e7
intr = torch.tensor([[90.0, 0, 6.5], [0, 90.0, 3.5], [0, 0, 1]]).repeat(V + 1, 1, 1)
disp_in = hashed((h1, w1), 82, 0.0, 0.002)
vol_ref, origin_ref = O.cost_volume(fm, poses, intr, D, incre, disp_in, True)
nhwc = (fm.permute(0, 2, 3, 1) / 8.0).reshape(V + 1, h1 * w1, C).contiguous().to(dev)
Pij = O.pij_matrices(poses, intr, [0] * V, [1, 2]).contiguous()
vol, origin = ops.cost_build(nhwc[0], nhwc[1:].contiguous(), Pij.to(dev), disp_in.reshape(-1).to(dev), D, incre, True, h1, w1, 3, fold=False)
d = (vol[..., :D].cpu() - vol_ref).abs()
print("max", d.max(), "sum", d.sum(), "ref sum", vol_ref.abs().sum())
idx = torch.nonzero(d > 1e-5)
print(idx[:40])
for v, p, k in idx[:10].tolist():
    hyp = (k - D // 2) * incre + float(origin_ref.reshape(-1)[p])
    print(v, p, k, "x,y", p % w1, p // w1, "u", p % w1 + 3600 * hyp, float(vol[v, p, k]), float(vol_ref[v, p, k]))
print(Pij)
