"""Multi-GPU use of the codec: blocks are independent (tables travel in-band), so a
batch is split into contiguous ranges, one per rank, and each rank runs the single-GPU pipeline on
its range.  There is no data-path collective; torch.distributed is used only for the barrier and
the max-over-ranks of the elapsed time when measuring."""
import os


def contiguous_partition(sizes, world):
    """Split blocks 0..n-1 into `world` contiguous ranges with near-equal total uncompressed bytes
    (greedy on the cumulative sum).  Returns [(lo, hi)] * world; ranges may be empty."""
    n = len(sizes)
    total = sum(sizes)
    bounds = [0]
    acc = 0
    i = 0
    for r in range(1, world):
        target = total * r / world
        while i < n and acc + sizes[i] / 2 <= target:
            acc += sizes[i]
            i += 1
        bounds.append(i)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


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
    kw = {}
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
