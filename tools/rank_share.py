#!/usr/bin/env python3
"""What ONE rank of a G-GPU sharded forward computes, measured on ONE GPU: the compute term of DESIGN.md section 6's
per-rank time model, from the kernels that ship today instead of round-2 timings.

  slab  (RAFT shard="slab"):  rank g's encoders (its source views, then reference + context), the cost volume on its row slab over all
        views, the GRU loop on the slab + halo, the strip pack / unpack launches - through slab.sharded_forward with an exchange object that
        owns rank g only and moves NOTHING between ranks (gathered feature slots / halo strips are filled with the rank's own data, so every
        kernel runs on realistic values and shapes);
  views (RAFT shard="views"): rank g's encoders (ceil(V / G) views + reference + context), the partial view-sum volume of its views,
        the replicated GRU loop - RAFT.forward with the group queries answered for (G, g) and the all-reduce skipped.

Communication is NOT included (that is the point: no second GPU is needed); DESIGN.md adds the modelled exchange terms.
usage: python tools/rank_share.py [--workload W] [--G 2,4,8] [--forwards 5] [--json out.json]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                               # noqa: E402
from cer_mvs_amd import RAFT, slab, dist as cdist                          # noqa: E402
from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene         # noqa: E402


class SoloExchange:
    """The exchange layer of slab.sharded_forward for ONE rank of G with the other ranks absent: same buffers and launches on this
    rank's side, no data from anyone else (their slots are filled with this rank's own data)."""

    def __init__(self, G, g):
        self.G, self.ranks, self._b = G, [g], {}

    def _buf(self, key, shape, t):
        if key not in self._b:
            self._b[key] = torch.empty(shape, device=t.device, dtype=t.dtype)
        return self._b[key]

    def gather_flat_async(self, tensors, tag="f"):
        t = tensors[0]
        out = self._buf((tag, tuple(t.shape)), (self.G,) + tuple(t.shape), t)
        out[self.ranks[0]].copy_(t)                        # (the rank's own contribution: the one copy a real all-gather also does locally)
        if not getattr(self, "_filled_" + tag, False):     # other ranks' slots: this rank's data, once (values only - never re-sent)
            for r in range(self.G):
                out[r].copy_(t)
            setattr(self, "_filled_" + tag, True)
        return out

    def wait_flat(self, handle):
        return [handle]

    def neighbor_exchange(self, bufs):
        half = bufs[0].numel() // 2
        g = self.ranks[0]
        rp = self._buf(("rp", half), (half,), bufs[0])
        rn = self._buf(("rn", half), (half,), bufs[0])
        if not getattr(self, "_nb_filled", False):
            rp.copy_(bufs[0][half:]); rn.copy_(bufs[0][:half]); self._nb_filled = True
        return [(rp if g > 0 else None, rn if g < self.G - 1 else None)]

    def all_gather_flat(self, tensors):
        t = tensors[0]
        out = self._buf(("agf", tuple(t.shape)), (self.G,) + tuple(t.shape), t)
        for r in range(self.G):
            out[r].copy_(t)
        return [out]

    def max_int(self, value, device):
        return int(value)


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dtu_1600x1184_v10_it32", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--G", default="2,4,8")
    ap.add_argument("--forwards", type=int, default=5)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    H, W, V, cascade = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda")
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=0)
    inputs = (images.to(dev), poses.to(dev), intr.to(dev))

    def make(**kw):
        m = RAFT(cascade=cascade, test_mode=True, gru_precision="s16f8", enc_precision="f6", cost_precision="x2", **kw)
        m.load_state_dict(fill_state_dict(m.state_dict(), seed=5))
        return m.to(dev).eval()

    res = {"workload": args.workload, "forwards_timed": args.forwards, "unit": "ms per depth map (one at a time, compute of one rank only)", "rows": []}
    with torch.no_grad():
        m1 = make()
        t1 = timed(lambda: m1(*inputs, scale=scale), args.forwards)
        res["single_gpu"] = t1
        print(f"single GPU, one at a time: {t1:.2f} ms")
        del m1
        h = H // 4
        orig = (cdist.group_info, cdist.reduce_volume, cdist.max_int, cdist.max_float)
        for G in [int(x) for x in args.G.split(",")]:
            for g in sorted({0, G // 2}):
                row = {"G": G, "rank": g}
                if slab.can_shard(h, G):
                    ms = make()
                    ex = SoloExchange(G, g)
                    r0, r1, e0, e1 = slab.slab_bounds(h, G, g)
                    row["slab_rows_owned"], row["slab_rows_computed"] = r1 - r0, e1 - e0
                    row["slab_ms"] = timed(lambda: slab.sharded_forward(ms, *inputs, scale, ex), args.forwards)
                    del ms, ex
                # view shard: the group queries answered for (G, g), the all-reduce skipped
                cdist.group_info = lambda group, G=G, g=g: (G, g) if group is not None else (1, 0)
                cdist.reduce_volume = lambda vol, group: vol
                cdist.max_int = lambda value, group, device: int(value)
                cdist.max_float = lambda value, group, device: float(value)
                try:
                    mv = make(view_group="solo", shard="views")
                    row["views_owned"] = len(cdist.local_views(V, "solo"))
                    row["views_ms"] = timed(lambda: mv(*inputs, scale=scale), args.forwards)
                    del mv
                finally:
                    cdist.group_info, cdist.reduce_volume, cdist.max_int, cdist.max_float = orig
                res["rows"].append(row)
                print(json.dumps(row))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
