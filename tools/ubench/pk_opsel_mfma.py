#!/usr/bin/env python3
"""Runs tools/ubench/pk_opsel_mfma.hip: every packed-fp32 operand-selection form on stream A while stream B runs (a) nothing, (b) an fp16 GEMM
(hipBLASLt: f16 MFMA), (c) an fp32 elementwise kernel.  Prints mismatch counts per form and companion.
build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -shared -fPIC tools/ubench/pk_opsel_mfma.hip -o tools/ubench/libpkopsel.so"""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libpkopsel.so"))
lib.pk_opsel_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
out = torch.zeros(2, dtype=torch.int64, device=dev)
a16 = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
a32 = torch.randn(8 * 1024 * 1024, device=dev)
abf = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
af32 = torch.randn(2048, 2048, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
names = {0: "add  src1 swapped", 1: "add  src0 swapped", 2: "mul  src1 swapped", 3: "fma  src1 swapped", 4: "add  src1 hi->both", 5: "add  src1 lo->both",
         6: "add  plain", 7: "fma  src0 hi->both", 8: "mul  src1 lo->both", 9: "mul  src0 hi->both", 11: "f16 add src1 swapped",
         12: "f16 mul src1 swapped", 13: "f16 add plain"}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
comps = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "fp16 matmul", "fp32 elementwise"]
forms = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(10))
for comp in comps:
    for form in forms:
        out.zero_()
        torch.cuda.synchronize()
        for r in range(rounds):
            with torch.cuda.stream(sB):
                for _ in range(12):
                    if comp == "fp16 matmul":
                        a16 @ a16
                    elif comp == "bf16 matmul":
                        abf @ abf
                    elif comp == "fp32 matmul":
                        af32 @ af32
                    elif comp == "fp32 elementwise":
                        a32.mul_(1.0001)
            with torch.cuda.stream(sA):
                for _ in range(10):
                    rc = lib.pk_opsel_launch(form, 2000, 768, out.data_ptr(), sA.cuda_stream)
                    assert rc == 0
            torch.cuda.synchronize()
        lo, hi = out.tolist()
        total = rounds * 10 * 768 * 256 * 2000
        print(f"companion {comp:16s} form {form:2d} ({names[form]:20s}): wrong low halves {lo:10d}, wrong high halves {hi:10d}  of {total:.2e} each", flush=True)
