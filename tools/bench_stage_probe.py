"""Stage timings with a synchronisation after every stage (Engine(profile=True)) for the bench workload;
development aid used to chase run-to-run differences.  usage: python tools/bench_stage_probe.py [anything]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tests_inputs import random_hopping
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.schmidt_utils import to_stopping_condition
L, chi = 1024, 512
C, N = slater.correlation_matrix(random_hopping(L, 0))
trunc = to_stopping_condition({"chi_max": chi})
for prof in (False, True):
    eng = Engine("cuda:0", profile=prof)
    d_C = torch.from_numpy(np.ascontiguousarray(C).reshape(-1)).to("cuda:0")
    for _ in range(3):
        eng.run(d_C, trunc, L // 2, L, download=False, threads=16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m = eng.run(d_C, trunc, L // 2, L, download=False, threads=16)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"argv={len(sys.argv)} profile={prof}: {dt*1e3:.1f} ms/step;", {k: round(v * 1e3, 1) for k, v in m.timings.items()}, flush=True)
