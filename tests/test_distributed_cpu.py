"""World-size-2 test of the N>1 path on CPU (gloo): view sharding + the per-stage all-reduce of the partial
view-sum volume reproduce the single-process view mean.  The per-rank partial volumes come from the oracle
(no GPU here); the partition / reduction / scaling code is the product's (cer-mvs_amd/dist.py)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import rel_l1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from cer_mvs_amd import dist as cdist
    from oracle import cer_oracle as O
    from test_oracle_golden import hashed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    h1, w1, C, D, incre = 6, 10, 64, 64, 0.0025 / 64
    fm = hashed((V + 1, C, h1, w1), 5, -2, 2)
    poses = torch.eye(4).repeat(V + 1, 1, 1)
    for v in range(1, V + 1):
        poses[v, 0, 3] = 25.0 * v
    intr = torch.tensor([[80.0, 0, 5.0], [0, 80.0, 3.0], [0, 0, 1]]).repeat(V + 1, 1, 1)
    disp = hashed((h1, w1), 6, 0.0, 0.002)
    views = cdist.local_views(V, dist.group.WORLD)
    if views:
        idx = [0] + views
        vol, origin = O.cost_volume(fm[idx], poses[idx], intr[idx], D, incre, disp, True)
        part = vol.sum(0).contiguous()
    else:
        part = torch.zeros(h1 * w1, D)
        origin = cdist.stage_origin(disp, D, incre, True)
    cdist.reduce_volume(part, dist.group.WORLD)
    mean = part / V
    full, origin_full = O.cost_volume(fm, poses, intr, D, incre, disp, True)
    ret[rank] = (views, rel_l1(mean, full.mean(0)), bool(torch.equal(origin, origin_full)))
    dist.barrier()
    dist.destroy_process_group()


def _run(V, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), V, ret), nprocs=world, join=True)
    return dict(ret)


def test_view_shards_sum_to_the_full_volume():
    out = _run(5)
    assert out[0][0] == [1, 3, 5] and out[1][0] == [2, 4]
    for r in (0, 1):
        assert out[r][1] < 1e-6 and out[r][2]


def test_view_shards_over_eight_ranks():
    """BASELINE configs[3] at G = 8: 10 views -> 2,2,1,1,1,1,1,1; the all-reduced partial sums give the full view mean on every rank."""
    out = _run(10, world=8)
    assert [len(out[r][0]) for r in range(8)] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert sorted(v for r in range(8) for v in out[r][0]) == list(range(1, 11))
    for r in range(8):
        assert out[r][1] < 1e-6 and out[r][2]


def test_more_ranks_than_views():
    out = _run(1)
    assert out[0][0] == [1] and out[1][0] == []
    for r in (0, 1):
        assert out[r][1] < 1e-6 and out[r][2]


def _agg_worker(rank, world, port, V, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from cer_mvs_amd import dist as cdist
    from test_oracle_golden import hashed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = hashed((V, 33, 5, 7), 91, -1.5, 3.0)
    views = cdist.local_views(V, dist.group.WORLD)
    local = frames[[v - 1 for v in views]] if views else frames[:0]
    got = cdist.aggregate_views(local, V, ["mean", "max", "std"], dist.group.WORLD)
    want = [frames.mean(0), frames.max(0).values, frames.std(0)]
    ret[rank] = [float((a - b).abs().max()) for a, b in zip(got, want)]
    dist.barrier()
    dist.destroy_process_group()


def test_view_aggregation_across_ranks_matches_torch():
    """The literal multi-GPU exchange: mean / max / std of the looked-up features over ALL views from per-rank
    partial views (core/update.py:101-107), incl. a rank without views."""
    for V in (5, 2, 1):
        if V == 1:
            continue                                   # torch.std of one view is NaN in the reference too
        ret = mp.Manager().dict()
        mp.spawn(_agg_worker, args=(2, _free_port(), V, ret), nprocs=2, join=True)
        for r in (0, 1):
            assert max(ret[r]) < 1e-5, (V, dict(ret))
    from cer_mvs_amd import dist as cdist
    from test_oracle_golden import hashed
    frames = hashed((4, 33, 3, 3), 92)
    one = cdist.aggregate_views(frames, 4, ["max", "mean"], None)           # single rank: plain torch, order mean, max
    assert torch.equal(one[0], frames.sum(0) / 4) and torch.equal(one[1], frames.max(0).values)


def test_partition_covers_every_view_once():
    from cer_mvs_amd import dist as cdist

    class FakeGroup:
        pass
    for V in (1, 7, 10, 15):
        for G in (1, 2, 4, 8):
            seen = []
            for g in range(G):
                seen += [v for v in range(1, V + 1) if (v - 1) % G == g]
            assert sorted(seen) == list(range(1, V + 1))
    assert cdist.local_views(10, None) == list(range(1, 11))


# ---------------------------------------------------------------------------------------------- row-slab sharding
def _fake_step(x, rows, w):
    """Stand-in for one update-block iteration: a vertical 15-tap filter + nonlinearity = receptive field of exactly 7 rows,
    zero padded at the edges of whatever tensor it is given (as the conv kernels do on a slab)."""
    import torch.nn.functional as F
    img = x.view(rows, w, -1).permute(2, 0, 1)[None]
    k = torch.linspace(0.2, 1.0, 15).view(1, 1, 15, 1).repeat(img.shape[1], 1, 1, 1)
    y = torch.tanh(F.conv2d(img, k, padding=(7, 0), groups=img.shape[1]) * 0.3 + 0.1 * img)
    return y[0].permute(1, 2, 0).reshape(rows * w, -1).contiguous()


def _torch_copy(pairs):
    """CPU stand-in for ops.copy_segments (cer_copy_segments_f32): same (src, dst) contiguous-range contract."""
    for src, dst in pairs:
        assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
        dst.view(-1).copy_(src.reshape(-1))


def _slab_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from cer_mvs_amd import slab
    from test_oracle_golden import hashed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    h, w, C = {2: 23, 4: 37, 8: 61}[world], 5, 5        # (7 * 5 * 6 floats per strip: not a multiple of 4 - the halves are padded)
    full = hashed((h * w, C), 77)
    ex = slab.DistExchange(dist.group.WORLD)
    r0, r1, e0, e1 = slab.slab_bounds(h, world, rank)
    x = full[e0 * w:e1 * w].clone()
    ref = full.clone()
    fulld = hashed((h * w,), 79)
    xd, refd = fulld[e0 * w:e1 * w].clone(), fulld.clone()
    buf = torch.zeros(2 * slab.strip_half(w, C))
    for _ in range(3):
        x = _fake_step(x, e1 - e0, w)
        ref = _fake_step(ref, h, w)
        xd = _fake_step(xd[:, None], e1 - e0, w)[:, 0].contiguous()
        refd = _fake_step(refd[:, None], h, w)[:, 0].contiguous()
        # the product's per-iteration exchange (slab.sharded_forward): pack -> neighbour point-to-point exchange -> refresh ...
        slab.pack_strips(x, xd, buf, w, r0, r1, e0, copy=_torch_copy)
        x2, xd2 = x.clone(), xd.clone()
        prev_half, next_half = ex.neighbor_exchange([buf])[0]
        slab.unpack_halves(x, xd, prev_half, next_half, w, rank, world, r0, r1, e0, e1, copy=_torch_copy)
        # ... and its all-gather form
        allbuf = ex.all_gather_flat([buf])[0]
        slab.unpack_halo(x2, xd2, allbuf, w, rank, world, r0, r1, e0, e1, copy=_torch_copy)
        assert torch.equal(x, x2) and torch.equal(xd, xd2)
        # and the list form (features / final gather use it)
        y = x.clone()
        strips = ex.all_gather([slab.border_strips(y, w, r0, r1, e0)])[0]
        slab.refresh_halo(y, strips, w, rank, world, r0, r1, e0, e1)
        assert torch.equal(x, y)
        # the persistent flat gather of the feature / disparity exchange
        flat = ex.wait_flat(ex.gather_flat_async([buf], tag="t"))[0]
        assert flat.shape[0] == world and torch.equal(flat[rank], buf) and torch.equal(flat, allbuf)
    own = x[(r0 - e0) * w:(r1 - e0) * w]
    ret[rank] = (float((own - ref[r0 * w:r1 * w]).abs().max()),
                 max(float((x - ref[e0 * w:e1 * w]).abs().max()), float((xd - refd[e0 * w:e1 * w]).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_slabs_with_halo_exchange_reproduce_the_full_image(world):
    """world = 4 / 8: middle ranks exchange with BOTH neighbours (2 sends + 2 receives in one batch_isend_irecv)."""
    ret = mp.Manager().dict()
    mp.spawn(_slab_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r][0] < 1e-6 and ret[r][1] < 1e-6, dict(ret)


def test_slab_bounds_cover_the_image():
    from cer_mvs_amd import slab
    for h in (16, 37, 120, 296, 384, 540):
        for G in (1, 2, 3, 4, 8):
            rows = []
            for g in range(G):
                r0, r1, e0, e1 = slab.slab_bounds(h, G, g)
                assert 0 <= e0 <= r0 < r1 <= e1 <= h and r0 - e0 <= slab.HALO and e1 - r1 <= slab.HALO
                rows += list(range(r0, r1))
            assert rows == list(range(h))
    assert slab.can_shard(296, 8) and not slab.can_shard(40, 8)


def test_local_exchange_simulation_matches_full_image():
    """The same bookkeeping with G simulated ranks in one process (what the GPU test of the real kernels uses)."""
    from cer_mvs_amd import slab
    from test_oracle_golden import hashed
    h, w, C, G = 40, 5, 3, 4
    full = hashed((h * w, C), 78)
    ex = slab.LocalExchange(G)
    b = [slab.slab_bounds(h, G, g) for g in range(G)]
    xs = [full[e0 * w:e1 * w].clone() for (_, _, e0, e1) in b]
    ref = full.clone()
    for _ in range(4):
        xs = [_fake_step(x, e1 - e0, w) for x, (_, _, e0, e1) in zip(xs, b)]
        ref = _fake_step(ref, h, w)
        got = ex.all_gather([slab.border_strips(x, w, r0, r1, e0) for x, (r0, r1, e0, e1) in zip(xs, b)])
        ys = [x.clone() for x in xs]
        for g in range(G):
            slab.refresh_halo(xs[g], got[g], w, g, G, *b[g])
        # flat pack / gather / refresh path on the same data (disp = channel 0 as a separate tensor)
        ds = [y[:, 0].contiguous() for y in ys]
        bufs = [torch.zeros(2 * slab.strip_half(w, C)) for _ in range(G)]
        for g, (r0, r1, e0, e1) in enumerate(b):
            slab.pack_strips(ys[g], ds[g], bufs[g], w, r0, r1, e0, copy=_torch_copy)
        flat = ex.all_gather_flat(bufs)
        nb = ex.neighbor_exchange(bufs)
        for g in range(G):
            y2, d2 = ys[g].clone(), ds[g].clone()
            slab.unpack_halo(ys[g], ds[g], flat[g], w, g, G, *b[g], copy=_torch_copy)
            slab.unpack_halves(y2, d2, nb[g][0], nb[g][1], w, g, G, *b[g], copy=_torch_copy)
            assert torch.equal(ys[g], xs[g]) and torch.equal(ds[g], xs[g][:, 0])
            assert torch.equal(y2, xs[g]) and torch.equal(d2, xs[g][:, 0])
    for g, (r0, r1, e0, e1) in enumerate(b):
        assert float((xs[g] - ref[e0 * w:e1 * w]).abs().max()) < 1e-6
