"""N>1 path of bench.py on CPU: two gloo ranks, each 'converting' its own stream; the job time is the MAX over
ranks, the job's frames the SUM (independent streams shard one per GPU - no collective on the data path)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dist.barrier()
    wall, frames = bench.reduce_job(0.5 + 0.25 * rank, 1600 + rank, torch.device("cpu"), True)
    q.put((rank, wall, frames))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduction_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, wall, frames in res:
        assert abs(wall - 0.75) < 1e-12        # max over ranks
        assert frames == 1600 + 1601           # sum over ranks
    assert round(res[0][2] / res[0][1], 1) == round(3201 / 0.75, 1)


def test_single_rank_reduction_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reduce_job(0.125, 320, torch.device("cpu"), False) == (0.125, 320)
