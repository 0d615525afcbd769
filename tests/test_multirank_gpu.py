"""Two real processes (torch.distributed, gloo, both on cuda:0) run the product's sharded RAFT.forward - every rank with the
HIP kernels on device tensors, the exchange through torch.distributed - and must reproduce the reference captures like the
single-process path.  (The GPU test boxes have one GPU; RCCL refuses two ranks on one device, so the transport here is gloo.
The collectives the product issues - all_gather / all_reduce on device tensors - are the same calls RCCL serves on a node.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, rel_l1

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


# Collectives of these tests time out after 5 minutes instead of gloo's default 30: one verification run of round 6 sat for 35 minutes in the FIRST
# two-process test on one box (the same suite had passed on three other boxes the same day; every later multi-process test of that run passed too).
import datetime
_PG_TIMEOUT = datetime.timedelta(seconds=300)


def _spawn(fn, args, nprocs):
    """mp.spawn with ONE retry on a fresh port: the transport of these tests (gloo between processes that share one GPU) is test infrastructure -
    the product's collectives run over RCCL - and a stuck rendezvous / host-staged copy must not be mistaken for a wrong result.  A second failure
    is a failure; the retry is reported."""
    try:
        mp.spawn(fn, args=args, nprocs=nprocs, join=True)
    except Exception as e:                                   # (ProcessRaisedException / ProcessExitedException)
        import warnings
        warnings.warn(f"multi-process test: first attempt failed ({type(e).__name__}: {str(e)[:200]}); retrying once")
        args = tuple(_free_port() if (isinstance(a, int) and 20000 < a < 65536 and i == 1) else a for i, a in enumerate(args))
        mp.spawn(fn, args=args, nprocs=nprocs, join=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shard, name, out_path, aggregation=None):
    import sys
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
    model = RAFT(cascade=cascade, test_mode=True, view_group=dist.group.WORLD, shard=shard)
    if aggregation is not None:
        from cer_mvs_amd.update import UpdateBlock
        model.update_block = UpdateBlock(cascade=model.cascade, dim_net=64, dim_inp=64, aggregation=aggregation)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    with torch.no_grad():
        out = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale)
    torch.cuda.synchronize()
    np.save(f"{out_path}.{rank}.npy", out.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard,name", [("slab", "e2e_cfg1"), ("views", "e2e_cfg1"), ("slab", "e2e_tiny"), ("views", "e2e_cfg2")])
def test_two_process_sharded_forward(dev, golden, tmp_path, shard, name):
    world = 2
    out_path = str(tmp_path / "disp")
    _spawn(_worker, (world, _free_port(), shard, name, out_path), world)
    ref = torch.from_numpy(golden(name)["disp"])
    outs = [torch.from_numpy(np.load(f"{out_path}.{r}.npy")) for r in range(world)]
    assert outs[0].shape == ref.shape
    errs = [rel_l1(o, ref) for o in outs]
    assert max(errs) < TOL, errs
    # every rank returns the same full disparity map (the host-staged gloo transport of this test does not promise bit-identical
    # sums on both ranks; over RCCL the reduced volume is bit-identical and so is everything after it)
    assert rel_l1(outs[0], outs[1]) < 1e-6, (rel_l1(outs[0], outs[1]), errs)


@pytest.mark.parametrize("shard", ["slab", "views"])
def test_four_process_sharded_forward(dev, golden, tmp_path, shard):
    """Four real processes: ranks 1 and 2 have BOTH neighbours (4 point-to-point operations in one batch per halo refresh), every
    rank owns views through the G > 2 partition (3 views over 4 ranks at e2e_tiny would leave one without: cfg1 has 2 views, so
    two ranks own none and contribute zeros / encode nothing), and the feature all-gather lands in the [G, vmax, ...] buffer the
    cost-volume kernel reads through its view -> block map."""
    world, name = 4, "e2e_cfg1"
    out_path = str(tmp_path / "disp")
    _spawn(_worker, (world, _free_port(), shard, name, out_path), world)
    ref = torch.from_numpy(golden(name)["disp"])
    outs = [torch.from_numpy(np.load(f"{out_path}.{r}.npy")) for r in range(world)]
    errs = [rel_l1(o, ref) for o in outs]
    assert outs[0].shape == ref.shape and max(errs) < TOL, errs
    assert max(rel_l1(outs[0], o) for o in outs[1:]) < 1e-6


def _stress_worker(rank, world, n, out_path):
    import sys
    sys.path.insert(0, REPO)
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = np.load(os.path.join(REPO, "tests", "golden", "e2e_cfg2.npz"))
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
    ref = torch.from_numpy(g["disp"]).to(dev).double()
    model = RAFT(cascade=cascade, test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    x = (images.to(dev), poses.to(dev), intr.to(dev))
    errs, differs, first = [], 0, None
    with torch.no_grad():
        for _ in range(n):
            o = model(*x, scale=scale)
            errs.append(float((o.double() - ref).abs().sum() / ref.abs().sum()))
            if first is None:
                first = o.clone()
            elif not torch.equal(o, first):
                differs += 1
    np.save(f"{out_path}.{rank}.npy", np.array([max(errs), differs, model.check_overflow(dev, raise_error=False)], dtype=np.float64))


def test_concurrent_processes_reproduce_the_capture_every_time(dev, tmp_path):
    """Guard for the corruption seen in round 2 (a removed lookup specialisation gave wrong features in 17 of 70 two-process runs):
    two INDEPENDENT processes share the GPU and run the bench workload 30 times each; every single output must match the reference
    capture (1e-4) and be bit-identical to the process's first output.  (tools/stress_parity.py is the open-ended form.)"""
    world, n = 2, 30
    out_path = str(tmp_path / "stress")
    _spawn(_stress_worker, (world, n, out_path), world)
    for r in range(world):
        worst, differs, overflow = np.load(f"{out_path}.{r}.npy")
        assert worst < TOL and differs == 0 and overflow == 0, (r, worst, differs, overflow)


def test_two_process_literal_forward_with_max_aggregation(dev, golden, tmp_path):
    """aggregation = [mean, max] has no view-mean fold: the literal forward shards the views and aggregates the looked-up
    features across ranks every GRU step (dist.aggregate_views: SUM and MAX all-reduces of [33, P]); it must equal the
    single-process literal forward."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.update import UpdateBlock
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    name, agg = "e2e_tiny", ("mean", "max")
    world = 2
    out_path = str(tmp_path / "disp")
    _spawn(_worker, (world, _free_port(), "views", name, out_path, agg), world)
    g = golden(name)
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    images, poses, intr, scale = synthetic_scene(int(g["H"]), int(g["W"]), int(g["V"]), seed=int(g["scene_seed"]))
    model = RAFT(cascade=cascade, test_mode=True)
    model.update_block = UpdateBlock(cascade=model.cascade, dim_net=64, dim_inp=64, aggregation=agg)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    with torch.no_grad():
        ref = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).cpu()
    outs = [torch.from_numpy(np.load(f"{out_path}.{r}.npy")) for r in range(world)]
    assert rel_l1(outs[0], outs[1]) < 1e-6 and rel_l1(outs[0], ref) < 1e-5 and rel_l1(outs[1], ref) < 1e-5, \
        (rel_l1(outs[0], outs[1]), rel_l1(outs[0], ref), rel_l1(outs[1], ref))


def _ragged_worker(rank, world, port, shard, H, W, V, out_path):
    import sys
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=31)
    model = RAFT(cascade=[(64, 64, 2), (-1, 320, 2)], test_mode=True, view_group=dist.group.WORLD, shard=shard)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=12))
    model = model.to(dev).eval()
    with torch.no_grad():
        out = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale)
    torch.cuda.synchronize()
    np.save(f"{out_path}.{rank}.npy", out.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard,V", [("slab", 3), ("views", 3), ("views", 1), ("slab", 1)])
def test_two_process_forward_at_ragged_size(dev, tmp_path, shard, V):
    """Uneven work per rank: 25 feature rows over 2 ranks (13 + 12, partial m-tile rows, odd width 33) and 3 views over 2 ranks
    (2 + 1) - or ONE view, so that a rank owns none (it contributes zeros / encodes nothing); every rank must reproduce the
    single-process forward."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    H, W, world = 100, 132, 2
    out_path = str(tmp_path / "disp")
    _spawn(_ragged_worker, (world, _free_port(), shard, H, W, V, out_path), world)
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=31)
    model = RAFT(cascade=[(64, 64, 2), (-1, 320, 2)], test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=12))
    model = model.to(dev).eval()
    with torch.no_grad():
        ref = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).cpu()
    for r in range(world):
        got = torch.from_numpy(np.load(f"{out_path}.{r}.npy"))
        assert got.shape == ref.shape and rel_l1(got, ref) < 1e-5, (r, rel_l1(got, ref))


def _worker_pipelined(rank, world, port, shard, name, out_path, streams):
    import sys
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.pipeline import DepthMapPipeline
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
    groups = [dist.new_group(ranks=list(range(world))) for _ in range(streams)]      # one communicator per depth map in flight
    model = RAFT(cascade=cascade, test_mode=True, view_group=groups[0], shard=shard)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    try:
        DepthMapPipeline(model, streams=streams)             # several in flight on ONE communicator: refused
        raise AssertionError("a sharded pipeline without per-replica groups must be refused")
    except ValueError:
        pass
    pipe = DepthMapPipeline(model, streams=streams, groups=groups)
    assert [m.view_group for m in pipe.models] == groups
    inputs = (images.to(dev), poses.to(dev), intr.to(dev))
    with torch.no_grad():
        handles = [pipe.submit(*inputs, scale) for _ in range(2 * streams + 1)]
        outs = [pipe.result(h_).cpu().numpy() for h_ in handles]
    torch.cuda.synchronize()
    np.save(f"{out_path}.{rank}.npy", np.stack(outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard", ["slab", "views"])
def test_two_process_pipelined_sharded_forward(dev, golden, tmp_path, shard):
    """a SHARDED model with two depth maps in flight - pipeline.DepthMapPipeline(groups=...): one process group
    (communicator) per replica, so collectives of different depth maps never interleave on one communicator.  Two real processes x
    two streams, five forwards of the cfg1 capture's input: every result of every rank matches the reference capture."""
    world, streams, name = 2, 2, "e2e_cfg1"
    out_path = str(tmp_path / "disp")
    _spawn(_worker_pipelined, (world, _free_port(), shard, name, out_path, streams), world)
    ref = torch.from_numpy(golden(name)["disp"])
    for r in range(world):
        outs = torch.from_numpy(np.load(f"{out_path}.{r}.npy"))
        assert outs.shape[0] == 2 * streams + 1
        for o in outs:
            assert rel_l1(o, ref) < TOL
