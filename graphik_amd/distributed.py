"""Multi-GPU sharding of an IK batch: one process per GPU, contiguous shards, no data-path
collective; ONE gather of the results at the end (torch.distributed: RCCL over xGMI on the GPU
box, gloo on CPU for tests).  IK goals are independent, so nothing else is exchanged.

    q, Y, info = solve_batch_sharded(graph, T_goals)        # every rank calls it with the full batch

is the multi-GPU twin of solvers.riemannian_solver.solve_batch (what ONE rank returns is what
solve_with_riemannian returns per goal, graphik/solvers/riemannian_solver.py:220-234: the joint
angles and the point matrix): rank r solves rows shard_range(B, r, world) on its GPU and the
per-problem results -- q [n], the statistics, on request the points Y [N*k] (~520 B
per problem at N = 18) -- are gathered on rank `dst` in one collective."""
import os

import numpy as np
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


_HAS_GATHER = {"nccl": True, "gloo": True}      # (RCCL and gloo implement gather: no probe, no extra collective)


def backend_has_gather():
    """Whether the process group's backend implements `gather` -- known for RCCL and gloo; for any other backend decided
    ONCE, by the same probe on every rank (a one-element gather at the first use, where all ranks are in step), never from the text of an
    exception raised in the middle of a job: a timeout or an asynchronous RCCL error mentions 'gather' too, and a rank
    that answers it by entering a different collective hangs the others."""
    be = dist.get_backend()
    if be not in _HAS_GATHER:
        rank, world = dist.get_rank(), dist.get_world_size()
        dev = torch.device("cuda", torch.cuda.current_device()) if be == "nccl" else torch.device("cpu")
        probe = torch.zeros(1, dtype=torch.float64, device=dev)
        try:
            dist.gather(probe, [torch.empty_like(probe) for _ in range(world)] if rank == 0 else None, dst=0)
            ok = 1.0
        except NotImplementedError:
            ok = 0.0
        except RuntimeError as e:
            if "does not support gather" not in str(e) and "not implemented" not in str(e).lower():
                raise
            ok = 0.0
        # every rank must come to the same answer (a refusal is raised before any communication, on all ranks alike)
        flag = torch.tensor([ok], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        _HAS_GATHER[be] = bool(flag.item() > 0.5)
    return _HAS_GATHER[be]


# what the last gather_rows() call of this process did: {"collective", "sent_bytes", "recv_bytes"} --
# a rank other than `dst` must receive nothing (ONE gather; tests assert recv_bytes == 0)
LAST_GATHER = {}


def gather_rows(local, total, dst=0):
    """Gather row-sharded tensors (shards as produced by shard_range) onto rank `dst`.
    Returns the [total, ...] tensor on dst, None elsewhere.  Single collective: `gather` to `dst`
    (ncclGather-equivalent under RCCL: every peer sends its rows over its own xGMI link to dst and
    receives nothing); `all_gather` only if the backend refuses `gather`."""
    if not dist.is_initialized():
        LAST_GATHER.update(collective="none", sent_bytes=0, recv_bytes=0)
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    collective = "gather" if backend_has_gather() else "all_gather"
    if collective == "gather":
        dist.gather(pad, bufs, dst=dst)          # (an error here is an error: no silent change of collective)
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
    row_bytes = pad.element_size() * int(np.prod(pad.shape))
    LAST_GATHER.update(collective=collective, sent_bytes=row_bytes if rank != dst or collective != "gather" else 0,
                       recv_bytes=0 if bufs is None else row_bytes * (world - 1))
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


# the per-problem statistics that travel with q (one float64 column each, in this order)
RESULT_STATS = ("pos_err", "rot_err", "f", "gradnorm", "iterations", "inner_total", "n_accept", "stop",
                "inner_executed")


def pack_results(res, with_Y=False, device=None):
    """One float64 row per problem of this rank's shard: q [n] | RESULT_STATS | (Y [N*k]).
    `res`: dict of tensors or arrays with keys "q", RESULT_STATS and (with_Y) "x".  A shard of zero
    rows packs to [0, width] (the widths come from the trailing dimensions, never from -1)."""
    def col(v):
        t = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
        t = t.to(torch.float64)
        if device is not None:
            t = t.to(device)
        return t.reshape(t.shape[0], int(np.prod(t.shape[1:])) if t.dim() > 1 else 1)
    cols = [col(res["q"])] + [col(res[k]) for k in RESULT_STATS]
    if with_Y:
        cols.append(col(res["x"]))
    return torch.cat(cols, dim=1).contiguous()


def empty_results(n, N=0, k=0, with_Y=False, device=None):
    """The [0, width] table of a rank whose shard is empty (more ranks than goals): it still enters
    the gather, it never touches the device solve."""
    width = n + len(RESULT_STATS) + (N * k if with_Y else 0)
    return torch.zeros((0, width), dtype=torch.float64, device=device)


def unpack_results(table, n, with_Y=False, point_shape=None):
    """Inverse of pack_results on the gathered [B, width] table -> (q [B,n], Y or None, info)."""
    a = table.cpu().numpy() if torch.is_tensor(table) else np.asarray(table)
    q = a[:, :n].copy()
    info = {k: a[:, n + i].copy() for i, k in enumerate(RESULT_STATS)}
    for k in ("iterations", "inner_total", "n_accept", "stop", "inner_executed"):
        info[k] = info[k].astype(np.int64)
    Y = None
    if with_Y:
        Y = a[:, n + len(RESULT_STATS):].copy()
        if point_shape is not None:
            Y = Y.reshape((len(a),) + tuple(point_shape))
    return q, Y, info


def result_row_bytes(n, N=0, k=0, with_Y=False):
    return 8 * (n + len(RESULT_STATS) + (N * k if with_Y else 0))


def solve_batch_sharded(graph, T_goals, use_limits=True, params=None, with_Y=False, dst=0, solve_fn=None):
    """solve_batch over all ranks of the process group (one rank per GPU; without a process group:
    this process alone).  EVERY rank passes the full batch T_goals [B, ...] (128 B per goal; anything
    with len() and row slicing -- an ndarray, a memory map, a lazy sequence: a rank only ever
    materialises ITS rows); rank r solves rows shard_range(B, r, world) and the results are gathered
    on rank `dst` in ONE collective of result_row_bytes() per problem.  Returns (q [B,n], Y [B,N,k]
    or None, info) on `dst`, (None, None, None) elsewhere.  A rank whose shard is empty (B < world)
    skips the solve and contributes zero rows.  `solve_fn(T_local) -> dict` replaces the device solve
    (tests of the rank / shard / gather logic on machines without a GPU)."""
    B = len(T_goals)
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_range(B, rank, world)
    T = np.asarray(T_goals[lo:hi], dtype=float)           # this rank's rows only
    n = graph.robot.n
    N, k = graph.number_of_nodes(), graph.dim
    cpu_group = dist.is_initialized() and dist.get_backend() != "nccl"
    on_gpu = not cpu_group and torch.cuda.is_available()
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    if hi == lo:
        table = empty_results(n, N, k, with_Y=with_Y, device=dev)
    elif solve_fn is not None:
        table = pack_results(solve_fn(T), with_Y=with_Y, device=dev)
    else:
        from .solvers.riemannian_solver import _problem_for, solve_batch
        prob = _problem_for(graph, use_limits, params, None)
        # The one place where the SIZE of a call selects arithmetic: planar graphs of at most 16 nodes run four problems
        # to a wavefront from 12 problems per CU on, one below (include/graphik_amd.h).  A rank sees only its shard, so
        # the rule is applied here to the GLOBAL batch and pinned (debug_flags 16384 / 8192): the gathered table then
        # does not depend on the world size.
        info = prob.template.info
        if info.get("problems_per_wave") == 4 and not (int((params or {}).get("debug_flags", 0)) & (8192 | 16384)):
            pin = 16384 if B >= 12 * int(info["n_cu"]) else 8192
            pinned = dict(params or {}, debug_flags=int((params or {}).get("debug_flags", 0)) | pin)
            prob = _problem_for(graph, use_limits, pinned, None)
        dev = torch.device("cpu") if cpu_group else prob.template.device
        if prob.device_pipeline:          # prepare -> solve -> recover on the device, results stay there
            res = prob.template.ik(torch.from_numpy(np.ascontiguousarray(T)).to(prob.template.device))
        else:
            q, Y, info = solve_batch(graph, T, use_limits=use_limits, params=params)
            res = dict(info, q=q, x=Y, f=info["f(x)"], inner_total=info["inner_iterations"],
                       n_accept=np.zeros(len(q)), inner_executed=info["inner_iterations"])
        table = pack_results(res, with_Y=with_Y, device=dev)
    table = gather_rows(table, B, dst=dst)
    if rank != dst:
        return None, None, None
    return unpack_results(table, n, with_Y=with_Y, point_shape=(N, k))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_vectors(vec, device):
    """A few doubles from every rank on every rank ([world][len(vec)] as lists): bookkeeping outside any timed region
    (bench.py's per-rank record), not part of the data path."""
    if not dist.is_initialized():
        return [[float(v) for v in vec]]
    t = torch.tensor([float(v) for v in vec], dtype=torch.float64, device=device)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return [[float(v) for v in b.cpu()] for b in bufs]


def sum_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
