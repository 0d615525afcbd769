#!/bin/bash
out=gpurun_out/r05d; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_conv_s16_gpu.py tests/test_determinism_gpu.py -x -q -k "lookup or end_to_end or fused_pyramid or corrblock or view_mean or slab or determin or pipeline" 2>&1 | tail -15 | tee $out/tests.txt
for cv in 0 1; do CER_COMPACT_VOLUME=$cv timeout 600 python bench.py --no-cpu-baseline --streams 1 2>/dev/null | grep "^{" > $out/bench_s1_cv$cv.json; done
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_s3.json
CER_MVS_LIB=$PWD/cer-mvs_amd/csrc/variants/libcermvs_w2.so timeout 600 python bench.py --no-cpu-baseline --streams 1 2>/dev/null | grep "^{" > $out/bench_s1_w2.json
CER_MVS_LIB=$PWD/cer-mvs_amd/csrc/variants/libcermvs_w2.so timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_s3_w2.json
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05d/bench_*.json")):
    try:
        d=json.loads(open(f).read()); k=d.get("kernels",{})
        print(f.split("/")[-1], round(d["value"],2), round(d["ms_per_step"],3), d.get("parity",{}).get("rel_l1_disp"), {n:round(v.get("avg_us",0),1) for n,v in k.items() if "lookup" in n or "delta_sum" in n or "cost" in n or "conv3x3" in n})
    except Exception as e: print(f, "ERR", e)
P
