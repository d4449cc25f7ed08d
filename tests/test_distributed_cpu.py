"""world_size-2 gloo test (CPU) of the multi-rank plumbing of bench.py: identical, contiguous,
exhaustive site shards on every rank and the max-over-ranks step time.  The compute itself
needs a GPU and is covered by tests/test_gpu_sweep.py::test_site_sharding_*."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_and_timing_reduce():
    code = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        L, oc = 1024, 512
        ranges = bench.shard_sites(L, oc, world)
        mine = ranges[rank]
        # every rank derives the same partition; ranges are contiguous and cover all sites
        gathered = [None] * world
        dist.all_gather_object(gathered, ranges)
        assert all(g == ranges for g in gathered)
        assert ranges[0][0] == 0 and ranges[-1][1] == L
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert abs(t.item() - 0.1 * world) < 1e-12
        n = torch.tensor([mine[1] - mine[0]])
        dist.all_reduce(n)
        assert n.item() == L
        dist.barrier()
        dist.destroy_process_group()
        print("ok", rank, mine)
    """ % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_shard_cost_balance():
    import sys
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench

    for world in (1, 2, 4, 8):
        r = bench.shard_sites(1024, 512, world)
        assert len(r) == world and sum(b - a for a, b in r) == 1024
        i = np.arange(1024)
        n = np.where(i < 512, i + 1, 1024 - i)
        w = 1.0 + 3.0 * (n / 512) ** 3
        loads = [w[a:b].sum() for a, b in r]
        assert max(loads) / (sum(loads) / world) < 1.15
