"""Multi-GPU sharding of an IK batch: one process per GPU, contiguous shards, no data-path
collective; one gather of the results at the end (torch.distributed: RCCL over xGMI on the GPU
box, gloo on CPU for tests).  IK goals are independent, so nothing else is exchanged."""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_process_group(backend=None):
    rank, local_rank, world = env_world()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # under torch.distributed.run
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total, rank, world):
    """Contiguous split: rank r owns [lo, hi).  Remainder goes to the first ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, total, dst=0):
    """Gather row-sharded tensors (shards as produced by shard_range) onto rank `dst`.
    Returns the [total, ...] tensor on dst, None elsewhere.  Single collective."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    if dist.get_backend() == "nccl":
        # all_gather is RCCL's best-supported path; the payload is a few hundred bytes/problem
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
    else:
        dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
