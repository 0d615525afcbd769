"""Multi-GPU plumbing of the hot path: source views shard across ranks, the view-sum cost
volume is the one exchange step (an all-reduce per cascade stage, RCCL over xGMI via torch.distributed),
everything after it is replicated.  Pure torch.distributed - the same code runs on gloo/CPU in the tests."""
import torch


def group_info(group):
    """(world_size, rank) of ``group``; (1, 0) when not distributed."""
    if group is None:
        return 1, 0
    import torch.distributed as dist
    return dist.get_world_size(group), dist.get_rank(group)


def local_views_for(num_views, G, g):
    """1-based source-view indices owned by rank g of G: v with (v-1) % G == g."""
    return [v for v in range(1, num_views + 1) if (v - 1) % G == g]


def local_views(num_views, group):
    """1-based source-view indices owned by this rank: v with (v-1) % G == g (cfg4: V=10 on G=2/4/8 ->
    5 / 3,3,2,2 / 2,2,1,1,1,1,1,1 views per rank).  May be empty when G > V."""
    G, g = group_info(group)
    return [v for v in range(1, num_views + 1) if (v - 1) % G == g]


def reduce_volume(vol, group):
    """In-place SUM all-reduce of the partial view-sum volume [P, row]; a no-op for a single rank."""
    G, _ = group_info(group)
    if G > 1:
        import torch.distributed as dist
        dist.all_reduce(vol, op=dist.ReduceOp.SUM, group=group)
    return vol


def stage_origin(disp, D, incre, shift):
    """Hypothesis origin (core/corr.py:59-62) for a rank that owns no view and therefore runs no build kernel."""
    if not shift:
        return disp.clone()
    lim = torch.tensor((D // 2) * incre, device=disp.device, dtype=torch.float32)
    return torch.where(disp < lim, lim, disp)


def aggregate_views(frames, num_views, aggregation, group):
    """View aggregation of the looked-up correlation features across ranks (core/update.py:101-107) - the literal multi-GPU form of
    SURVEY.md 8(e): ``frames`` [V_local, K, ...] holds this rank's views (V_local may be 0), every rank gets the list of
    aggregated [K, ...] tensors in the order mean, max, std.  mean: one SUM all-reduce of the local sum; max: one MAX all-reduce;
    std (torch.std: unbiased): a second SUM all-reduce of the squared deviations from the global mean."""
    G, _ = group_info(group)
    if frames.shape[0]:
        local_sum = frames.sum(0)
        shape, dev, dt = frames.shape[1:], frames.device, frames.dtype
    else:
        shape, dev, dt = frames.shape[1:], frames.device, frames.dtype
        local_sum = torch.zeros(shape, device=dev, dtype=dt)
    out = []
    need_mean = "mean" in aggregation or "std" in aggregation
    mean = None
    if need_mean:
        mean = local_sum.clone()
        if G > 1:
            import torch.distributed as dist
            dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=group)
        mean = mean / num_views
    if "mean" in aggregation:
        out.append(mean)
    if "max" in aggregation:
        mx = frames.max(0).values if frames.shape[0] else torch.full(shape, float("-inf"), device=dev, dtype=dt)
        if G > 1:
            import torch.distributed as dist
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        out.append(mx)
    if "std" in aggregation:
        ss = ((frames - mean) ** 2).sum(0) if frames.shape[0] else torch.zeros(shape, device=dev, dtype=dt)
        if G > 1:
            import torch.distributed as dist
            dist.all_reduce(ss, op=dist.ReduceOp.SUM, group=group)
        out.append(torch.sqrt(ss / (num_views - 1)))
    return out


def max_int(value, group, device):
    """MAX of a host integer over the ranks of ``group`` (one tiny all-reduce: the overflow flag under overflow_policy="raise")."""
    import torch.distributed as dist
    if group is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(value)
    on_dev = dist.get_backend(group) == "nccl"
    t = torch.tensor([int(value)], dtype=torch.int32, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def max_float(value, group, device):
    """MAX of a host float over the ranks of ``group`` (the calibration figure of gru_precision="auto": every rank takes the same decision)."""
    import torch.distributed as dist
    if group is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    on_dev = dist.get_backend(group) == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
