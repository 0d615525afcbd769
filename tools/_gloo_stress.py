import os, sys, torch
import torch.distributed as dist
import torch.multiprocessing as mp

def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    bad = 0
    n = 19200 * 112
    for it in range(60):
        gens = [torch.Generator(device="cpu").manual_seed(1000 * it + r) for r in range(world)]
        parts = [torch.randn(n, generator=g) for g in gens]
        want = (parts[0] + parts[1]).to(dev)
        mine = parts[rank].to(dev)
        y = torch.zeros(n, device=dev)
        y = y + mine * 2.0            # some device work right before the collective
        y = y - mine
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        z = y * 1.0                   # device work right after
        torch.cuda.synchronize()
        if not torch.equal(z, want):
            bad += 1
    ret[rank] = bad
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(2, 29561, ret), nprocs=2, join=True)
    print("mismatching all_reduce results per rank (of 60):", dict(ret))
