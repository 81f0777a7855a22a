"""N>1 path on CPU: two gloo ranks split a batch with contiguous_partition, each codes its own
range (the CPU oracle stands in for the device codec — this is a test of the sharding and of the
barrier / max-over-ranks plumbing that bench.py uses, not of the kernels), and the gathered result
must equal the unsharded one."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen
from htscodecs_amd import shard

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _blocks():
    names = ["q4", "q8", "q40+dir"]
    sizes = [3000, 70000, 12000, 500, 65536, 20000, 1, 40000, 9000]
    return [datagen.tile(names[b % 3], sizes[b % len(sizes)], b).tobytes() for b in range(23)]


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    import cpu_libs
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d = shard.init("gloo")
    blocks = _blocks()
    lo, hi = shard.contiguous_partition([len(b) for b in blocks], world)[rank]
    orc = cpu_libs.oracle()
    mine = [orc.compress(b, 1) for b in blocks[lo:hi]]
    d.barrier()
    sizes = shard.gather_sizes(d, [len(c) for c in mine])
    slowest = shard.max_over_ranks(d, 1.0 + rank)
    if rank == 0:
        q.put((sizes, slowest))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_sharding_matches_single():
    import cpu_libs
    orc = cpu_libs.oracle()
    blocks = _blocks()
    want = [len(orc.compress(b, 1)) for b in blocks]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    sizes, slowest = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sizes == want
    assert slowest == 2.0


def test_partition_properties():
    for world in (1, 2, 3, 8):
        for sizes in ([5] * 16, [1, 100, 1, 1, 50, 7], [10], [], [3, 3, 3]):
            parts = shard.contiguous_partition(sizes, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == len(sizes)
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and a <= b
    parts = shard.contiguous_partition([1 << 20] * 16, 8)
    assert all(hi - lo == 2 for lo, hi in parts)


def test_partition_is_the_librarys():
    """shard.contiguous_partition is rans4x16_hip_partition (host arithmetic inside librans4x16_hip.so, the code the
    C-level *_batch_multi calls use): midpoint rule, equal weights with NULL, bad arguments refused."""
    import ctypes as C
    import htscodecs_amd
    L = htscodecs_amd.load()
    b = (C.c_int * 4)()
    w = (C.c_uint * 6)(1, 100, 1, 1, 50, 7)
    assert L.rans4x16_hip_partition(6, w, 3, b) == 0
    assert list(b) == [0, 2, 4, 6]          # targets 53.3 / 106.7; midpoints 0.5, 51 | 101.5, 102.5 | 128, 156.5
    assert L.rans4x16_hip_partition(10, None, 3, b) == 0 and list(b) == [0, 3, 7, 10]
    assert L.rans4x16_hip_partition(-1, None, 3, b) == -1
    assert L.rans4x16_hip_partition(5, None, 0, b) == -1
    assert shard.uniform_share(8 * 15360, 8, 3) == (3 * 15360, 4 * 15360)


def test_bench_refuses_more_ranks_than_gpus():
    """bench.py --gpus 2 on a box with fewer GPUs says so instead of running a 1-rank job (the flag used
    to be parsed and dropped)."""
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "R4X16_OVERSUBSCRIBE")}
    if torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in r.stderr
