#!/usr/bin/env python3
"""A second GPU process that does almost nothing: tiny elementwise kernels in a loop for N seconds (does time-slicing alone trigger the failure?)."""
import sys, time, torch
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = sys.argv[2] if len(sys.argv) > 2 else "tiny"
x = torch.zeros(64 if mode == "tiny" else 64 * 1024 * 1024, device="cuda")
n = 0
while time.time() < t_end:
    for _ in range(200):
        x.add_(1.0)
    torch.cuda.synchronize()
    n += 200
print("spinner launches", n)
