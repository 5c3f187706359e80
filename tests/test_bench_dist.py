"""N>1 path of bench.py on CPU (gloo, world size 2): the REAL control flow of bench.main - self-spawn for `--gpus N`, rank
count check, barrier, max-over-ranks time, sum-over-ranks frames, per-rank gather - with a stub workload in place of the
converter (the HIP path has no CPU form).  Independent streams shard one per GPU: no collective on the data path."""
import json
import os
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=e, cwd=ROOT)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_by_itself():
    """`python bench.py --gpus 2` (no torchrun around it) must start two ranks and report n_gpus 2 - an ignored --gpus fails here."""
    r = _run([sys.executable, BENCH, "--gpus", "2", "--config", "stub", "--backend", "gloo", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2
    assert len(line["per_rank"]) == 2 and [p["rank"] for p in line["per_rank"]] == [0, 1]
    # rank 1's steps take twice as long: the job time is the slower rank's, the job's frames are both ranks'
    frames = 2 * 5 * 4
    assert abs(line["value"] - frames / (line["ms_per_step"] * 5e-3)) / line["value"] < 1e-3
    assert line["per_rank"][0]["value"] > line["per_rank"][1]["value"] * 1.3
    assert line["value"] < line["per_rank"][0]["value"] + line["per_rank"][1]["value"]
    assert line["value"] <= 2 * line["per_rank"][1]["value"] * 1.05
    assert "cpu_baseline" not in line           # rank 0 at N = 1 only


def test_under_torchrun_as_the_driver_launches_it():
    port = 29600 + (os.getpid() % 2000)
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), BENCH, "--gpus", "2", "--config", "stub", "--backend", "gloo", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 2


def test_rank_count_mismatch_is_an_error():
    """WORLD_SIZE says 1 rank but --gpus says 2: refuse instead of printing a 1-GPU number as a 2-GPU one."""
    r = _run([sys.executable, BENCH, "--gpus", "2", "--config", "stub", "--backend", "gloo"],
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dist.barrier()
    wall, frames = bench.reduce_job(0.5 + 0.25 * rank, 1600 + rank, torch.device("cpu"), True)
    per = bench.gather_ranks([10.0 + rank, 20.0 + rank], torch.device("cpu"), True, world)
    q.put((rank, wall, frames, per))
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
    for rank, wall, frames, per in res:
        assert abs(wall - 0.75) < 1e-12        # max over ranks
        assert frames == 1600 + 1601           # sum over ranks
        assert per == [[10.0, 20.0], [11.0, 21.0]]
    assert round(res[0][2] / res[0][1], 1) == round(3201 / 0.75, 1)


def test_single_rank_reduction_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reduce_job(0.125, 320, torch.device("cpu"), False) == (0.125, 320)
    assert bench.gather_ranks([1.0, 2.0], torch.device("cpu"), False, 1) == [[1.0, 2.0]]
