"""BASELINE config 4: Majorana chain (src/examples/iMPS_pfaffian.py:7-11) / random BdG, Pfaffian -> MPS."""
import argparse, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
from make_golden_pfaffian import random_majorana_H, kitaev_majorana_H
from temfpy_amd import pfaffian
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=512); ap.add_argument("--chi", type=int, default=256)
ap.add_argument("--random", action="store_true"); ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
os.environ.setdefault("TMF_PROFILE", "1")
H = random_majorana_H(a.L, 0) if a.random else kitaev_majorana_H(a.L, 1.5j, 1j)
C = pfaffian.correlation_matrix(H, "M->M")
for r in range(a.reps):
    t0 = time.perf_counter()
    mps = pfaffian.C_to_MPS(C, {"chi_max": a.chi}, basis="M")
    dt = time.perf_counter() - t0
    S = mps.entanglement_entropy(all_bonds=True)
    print(f"rep {r}: L={a.L} chi={a.chi} wall={dt:.3f}s -> {a.L/dt:.1f} sites/s S(centre)={S[a.L//2]:.9f} max chi={max(mps.chi)} max k={max(b.k for b in mps.bonds)}")
    print("    info:", mps.info)
    for k, v in mps.timings.items():
        print(f"    {k:22s} {v*1e3:10.1f} ms")
