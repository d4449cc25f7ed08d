"""Run one Slater -> MPS conversion on the GPU with per-stage timings (development aid)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests_inputs import random_hopping  # noqa: E402
from temfpy_amd import slater  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=1024)
ap.add_argument("--chi", type=int, default=512)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
os.environ.setdefault("TMF_PROFILE", "1")
C, N = slater.correlation_matrix(random_hopping(a.L, 0))
for r in range(a.reps):
    t0 = time.perf_counter()
    mps = slater.C_to_MPS(C, {"chi_max": a.chi}, as_tenpy=False)
    dt = time.perf_counter() - t0
    S = mps.entanglement_entropy(all_bonds=True)
    print(f"rep {r}: L={a.L} chi={a.chi} N={N} wall={dt:.3f}s -> {a.L/dt:.1f} sites/s  S(centre)={S[a.L//2]:.9f} max chi={max(mps.chi)}")
    for k, v in mps.timings.items():
        print(f"    {k:22s} {v*1e3:10.1f} ms")
