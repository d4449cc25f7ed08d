"""Host -> host time of one rank's site range of the benchmark chain, conversions pipelined as bench.py does (the download of
conversion k under conversion k + 1, consecutive conversions on two alternating contexts like multi_gpu.ShardGroup; EXP_CONTEXTS=1:
one), with the engine's stage timers: what the slowest rank of a sharded run costs.
usage: python tools/shard_host_cost.py lo:hi [lo:hi ...]      (default: the ranges shard_sites gives for 8 ranks)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.multi_gpu import shard_sites
from temfpy_amd.schmidt_utils import to_stopping_condition
import torch
L = 1024
C, _ = slater.correlation_matrix(random_hopping(L, 0))
tr = to_stopping_condition({"chi_max": 512})
ranges = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or shard_sites(L, L // 2, 8)[:4]
engs = [Engine("cuda:0") for _ in range(int(os.environ.get("EXP_CONTEXTS", 2)))]
ht = int(os.environ.get("EXP_THREADS", 16))
K = 40
for rng in ranges:
    res, turn = [], [0]
    def step():
        eng = engs[turn[0] % len(engs)]
        turn[0] += 1
        res.append(eng.run(C, tr, L // 2, L, download="async", threads=ht, site_range=rng))
        if len(res) > 1:
            res.pop(0).wait()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    while res:
        res.pop(0).wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    tm = {k: round(v * 1e3, 2) for k, v in engs[0].timings.items() if v * 1e3 >= 0.05}
    print(f"range {rng} ({rng[1] - rng[0]} sites): {dt * 1e3:6.2f} ms host -> host per conversion; stages {tm}", flush=True)
