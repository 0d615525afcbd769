#!/usr/bin/env python3
"""One process, S depth maps in flight (DepthMapPipeline), N forwards, every output compared with the first: does co-residency with this
program's own kernels (no second process) trigger the experimental lookup's failure?  usage: repro_streams.py <gru_precision> [streams] [forwards]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cer_mvs_amd import RAFT
from cer_mvs_amd.pipeline import DepthMapPipeline
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
prec = sys.argv[1]; S = int(sys.argv[2]) if len(sys.argv) > 2 else 3; n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
dev = torch.device("cuda")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "e2e_cfg2.npz"))
H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
casc = [tuple(int(x) for x in c) for c in g["cascade"]]
images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
enc = os.environ.get('RS_ENCODER', 'hip')
model = RAFT(cascade=casc, test_mode=True, gru_precision=prec, encoder_backend=enc)
if os.environ.get('RS_WALK') == '1':
    from cer_mvs_amd import _lib as L
    L.load().cer_cost_build_algo(1)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
model = model.to(dev).eval()
x = (images.to(dev), poses.to(dev), intr.to(dev))
with torch.no_grad():
    ref = model(*x, scale=scale).clone()
    pipe = DepthMapPipeline(model, streams=S)
    outs = list(pipe.map([(x[0], x[1], x[2], scale)] * n))
    bad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
print(f"gru_precision={prec} encoder={enc} walk={os.environ.get('RS_WALK', '0')} engine={os.environ.get('CER_ENC_ENGINE', 'pc')} streams={S}: {bad} of {n} forwards differ from the one-at-a-time result")
