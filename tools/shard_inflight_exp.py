"""Experiment: one rank's share of the benchmark chain (world, rank from the command line), host C in -> host tensors out,
with 1, 2 or 3 conversions in flight on as many contexts (launch streams).  A shard of an 8-way split is a chain of short
latency-bound launches: does the GPU overlap the chains of consecutive conversions?
usage: python tools/shard_inflight_exp.py <world> <rank>"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.multi_gpu import shard_sites
from temfpy_amd.schmidt_utils import to_stopping_condition
import torch
world, rank = int(sys.argv[1]), int(sys.argv[2])
L = 1024
C, _ = slater.correlation_matrix(random_hopping(L, 0))
tr = to_stopping_condition({"chi_max": 512})
rng = shard_sites(L, L // 2, world)[rank]
K = 48
print("range", rng, flush=True)
DEPTH = int(os.environ.get('EXP_DEPTH', 0))
for nctx, nthr in ((1, 1), (2, 1)) if DEPTH else ((1, 1), (2, 1), (3, 1), (2, 2), (3, 3)):
    engs = [Engine("cuda:0") for _ in range(nctx)]
    ht = int(os.environ.get('EXP_THREADS', 0)) or max(2, 16 // max(nthr, 1))

    def work(es, n):
        res = []
        for k in range(n):
            res.append(es[k % len(es)].run(C, tr, L // 2, L, download="async", threads=ht, site_range=rng))
            if len(res) > (DEPTH or len(es)):
                res.pop(0).wait()
        for r in res:
            r.wait()
    groups = [engs] if nthr == 1 else [[e] for e in engs]
    for g in groups:
        work(g, 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(g, K // len(groups))) for g in groups]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (K // len(groups) * len(groups))
    print(f"contexts={nctx} host threads={nthr}: {dt*1e3:6.2f} ms per conversion of the shard -> {L/dt:8.0f} sites/s if every rank did this", flush=True)
    del engs
