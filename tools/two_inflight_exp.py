"""Experiment: throughput with TWO conversions in flight on one GPU (two contexts = two launch streams, two Python
threads; the ctypes calls release the GIL), against one at a time."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.schmidt_utils import to_stopping_condition
import torch
C, _ = slater.correlation_matrix(random_hopping(1024, 0))
tr = to_stopping_condition({"chi_max": 512})
K = 24
for mode in (("async",) if os.environ.get("EXP_ASYNC_ONLY") else ("async", False)):
    for nthr, ht in (((1, 32), (2, 16)) if os.environ.get("EXP_ASYNC_ONLY") else ((1, 32), (2, 16), (2, 32))):
        engs = [Engine("cuda:0") for _ in range(nthr)]
        last = {}
        def work(e, n, ht=ht):
            res = []
            for _ in range(n):
                res.append(e.run(C, tr, 512, 1024, download=mode, threads=ht))
                last[id(e)] = res[-1].timings
                if len(res) > 2:
                    res.pop(0).wait()
            for r in res:
                r.wait()
        for e in engs:
            work(e, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(e, K // nthr)) for e in engs]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (K // nthr * nthr)
        print(f"download={mode!s:6s} contexts={nthr} host_threads={ht}: {dt*1e3:6.2f} ms per conversion -> {1024/dt:8.0f} sites/s", flush=True)
        if nthr > 1 and mode == "async":
            t = list(last.values())[0]
            print("      stages:", {k: round(v * 1e3, 1) for k, v in t.items() if v > 2e-4}, flush=True)
        del engs
