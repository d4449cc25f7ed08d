"""Run one Slater -> MPS conversion on the GPU with per-stage timings (development aid)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests_inputs import random_hopping, uniform_chain  # noqa: E402
from temfpy_amd import slater  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=1024)
ap.add_argument("--chi", type=int, default=512)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--chain", action="store_true", help="uniform chain (real) instead of random complex hopping")
ap.add_argument("--spinful", default=None)
ap.add_argument("--check", action="store_true", help="compare with the CPU oracle (small sizes)")
a = ap.parse_args()
os.environ.setdefault("TMF_PROFILE", "1")
C, N = slater.correlation_matrix(uniform_chain(a.L) if a.chain else random_hopping(a.L, 0))
for r in range(a.reps):
    t0 = time.perf_counter()
    mps = slater.C_to_MPS(C, {"chi_max": a.chi}, as_tenpy=False, spinful=a.spinful)
    dt = time.perf_counter() - t0
    S = mps.entanglement_entropy(all_bonds=True)
    Lm = mps.L
    print(f"rep {r}: sites={Lm} -> {Lm/dt:.1f} sites/s  max k={max(len(b.e) for b in mps.bonds)}")
    print(f"rep {r}: L={a.L} chi={a.chi} N={N} wall={dt:.3f}s -> {a.L/dt:.1f} sites/s  S(centre)={S[mps.L//2]:.9f} max chi={max(mps.chi)}")
    for k, v in mps.timings.items():
        print(f"    {k:22s} {v*1e3:10.1f} ms")

if a.check:
    from oracle import slater_oracle as orc
    t0 = time.perf_counter()
    cuts, sites = orc.c_to_mps(C, {"chi_max": a.chi}, spinful=a.spinful)
    print(f"oracle: {time.perf_counter()-t0:.2f}s")
    dS = np.abs(orc.entropies(cuts) - mps.entanglement_entropy(all_bonds=True)).max()
    same = all(m.chi == len(c.lam) and sorted(map(bytes, m.sets)) == sorted(map(bytes, c.sets)) for m, c in zip(mps.bonds, cuts))
    oc = mps.L // 2
    T1, T2 = orc.dense_tensors(cuts, sites), mps.dense_tensors()
    ov = abs(orc.mps_overlap(T1, cuts[oc].lam, T2, mps.lam[oc], oc)) / np.sqrt(abs(orc.mps_overlap(T1, cuts[oc].lam, T1, cuts[oc].lam, oc)) * abs(orc.mps_overlap(T2, mps.lam[oc], T2, mps.lam[oc], oc)))
    print(f"check: same kept subsets={same} max|dS|={dS:.2e} 1-overlap={1-ov:.2e}")
