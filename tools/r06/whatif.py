#!/usr/bin/env python3
"""What-if at the bench workload: ms per depth map, one at a time and with three in flight, with ONE component's launches turned into no-ops
(results are garbage; the timing says what that component costs in each regime - the most any optimisation of it could buy).  Round 2's
tools/archive/exp_whatif.py did this with cached results for the cost volume and the encoders; this form skips the library calls themselves, so it also
covers the update-block convolutions and the lookup.  usage: python tools/r06/whatif.py [--gru-precision s16f8]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                               # noqa: E402
from cer_mvs_amd import RAFT, _lib as L                                    # noqa: E402
from cer_mvs_amd.pipeline import DepthMapPipeline                          # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gru-precision", default="s16f8")
ap.add_argument("--enc-precision", default="f6", help="the shipped auto form on the bench weights is s16f8+e6: encoders in the FP6-correction form")
ap.add_argument("--forwards", type=int, default=12)
args = ap.parse_args()
H, W, V, cascade = bench.WORKLOADS["dtu_1600x1184_v10_it32"]
dev = torch.device("cuda")
images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
x = (images.to(dev), poses.to(dev), intr.to(dev))


class Skip:
    def __init__(self, lib, pred):
        self._lib, self._pred = lib, pred

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if self._pred(name) and not name.endswith(("_pack", "_size", "_scale", "_supported", "_workspace")):
            return lambda *a: 0
        return fn


def measure(label, pred):
    L._recorder = Skip(L.load() if L._recorder is None else L._lib, pred) if pred else None
    out = []
    try:
        for S in (1, 3):
            model = RAFT(cascade=cascade, test_mode=True, gru_precision=args.gru_precision, enc_precision=args.enc_precision, cost_precision="x2")   # (the shipped auto form on the bench weights: s16f8+e6+c2)
            model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
            model = model.to(dev).eval()
            model.overflow_policy = "ignore"
            pipe = DepthMapPipeline(model, streams=S)
            for m in pipe.models:
                m.overflow_policy = "ignore"
            with torch.no_grad():
                for _ in range(2 * S):
                    pipe.result(pipe.submit(*x, scale), wait_on_host=False)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for st in pipe.streams:
                    st.wait_event(e0)
                for _ in range(args.forwards):
                    pipe.submit(*x, scale)
                for st in pipe.streams:
                    torch.cuda.current_stream().wait_stream(st)
                e1.record()
                torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / args.forwards)
            pipe.close()
    finally:
        L._recorder = None
    print(f"{label:44s} one at a time {out[0]:7.3f} ms   three in flight {out[1]:7.3f} ms", flush=True)
    return out


base = measure("baseline", None)
for label, pred in (("update-block convolutions free", lambda n: n == "cer_conv3x3_s16"),
                    ("lookup free", lambda n: n == "cer_lookup_encode_f32"),
                    ("cost volume free", lambda n: n.startswith("cer_cost_lines") or n == "cer_cost_build_f32" or n == "cer_feat_split_f16"),
                    ("encoders free", lambda n: n.startswith("cer_enc_")),
                    ("everything above free (host + the rest)", lambda n: n in ("cer_conv3x3_s16", "cer_lookup_encode_f32", "cer_cost_build_f32", "cer_feat_split_f16")
                     or n.startswith(("cer_cost_lines", "cer_enc_")))):
    o = measure(label, pred)
    print(f"{'':44s}               {o[0] - base[0]:+7.3f} ms                   {o[1] - base[1]:+7.3f} ms", flush=True)
