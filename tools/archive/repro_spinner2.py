#!/usr/bin/env python3
"""Companion processes of one kind each: 'matmul' (fp16 GEMMs: MFMA), 'valu' (fp32 transcendental elementwise), 'lds' (sort: LDS heavy)."""
import sys, time, torch
t_end = time.time() + float(sys.argv[1])
mode = sys.argv[2]
dev = "cuda"
if mode == "matmul":
    a = torch.randn(4096, 4096, device=dev, dtype=torch.float16); b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
elif mode == "matmul32":
    a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
else:
    a = torch.randn(32 * 1024 * 1024, device=dev)
n = 0
while time.time() < t_end:
    for _ in range(20):
        if mode.startswith("matmul"):
            c = a @ b
        elif mode == "valu":
            c = torch.sin(a) * torch.cos(a) + torch.exp(-a * a)
        else:
            c = torch.sort(a.view(-1, 1024), dim=1)[0]
    torch.cuda.synchronize(); n += 20
print("companion", mode, n)
