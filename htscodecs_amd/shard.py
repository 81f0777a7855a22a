"""Multi-GPU use of the codec: blocks are independent (tables travel in-band), so a
batch is split into contiguous ranges, one per rank, and each rank runs the single-GPU pipeline on
its range.  There is no data-path collective; torch.distributed is used only for the barrier and
the max-over-ranks of the elapsed time when measuring."""
import os


def contiguous_partition(sizes, world):
    """Split blocks 0..n-1 into `world` contiguous ranges with near-equal total uncompressed bytes.
    The split itself is the library's (rans4x16_hip_partition, pure host arithmetic: it needs no GPU), the
    same code the C-level multi-device calls use.  Returns [(lo, hi)] * world; ranges may be empty."""
    import ctypes as C
    from . import lib as _lib
    n = len(sizes)
    w = (C.c_uint * max(n, 1))(*[int(x) for x in sizes])
    bounds = (C.c_int * (world + 1))()
    if _lib.load().rans4x16_hip_partition(n, w, world, bounds) != 0:
        raise ValueError("rans4x16_hip_partition: bad arguments")
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def uniform_share(nblocks_total, world, rank):
    """Range of rank `rank` when all blocks have the same size (weights NULL in the C call)."""
    import ctypes as C
    from . import lib as _lib
    bounds = (C.c_int * (world + 1))()
    if _lib.load().rans4x16_hip_partition(int(nblocks_total), None, world, bounds) != 0:
        raise ValueError("rans4x16_hip_partition: bad arguments")
    return bounds[rank], bounds[rank + 1]


def env_rank():
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def init(backend, device_id=None):
    """Join the process group given by the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank()
    if world == 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import datetime
    kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("R4X16_DIST_TIMEOUT_S", 600)))}   # a dead peer ends the job
    if backend == "nccl" and device_id is not None:
        kw["device_id"] = device_id
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(dist, value, device=None):
    """MAX-reduce a Python float over the ranks (identity without a process group)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_sizes(dist, local_sizes):
    """All ranks learn every rank's per-block result sizes, in block order."""
    if dist is None:
        return list(local_sizes)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local_sizes))
    return [s for part in out for s in part]


def gather_over_ranks(dist, value, device=None):
    """Every rank's Python float, in rank order (a one-element list without a process group)."""
    if dist is None:
        return [float(value)]
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
