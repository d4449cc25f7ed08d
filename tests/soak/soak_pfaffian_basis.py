"""Randomised soak of pfaffian.C_to_MPS across input bases: the same state handed over in the Majorana and in the complex-fermion
basis gives the same Schmidt values on every bond.  Development aid."""
import os
import sys, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from temfpy_amd import pfaffian
bad = 0
for seed in range(300):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(2, 15)); rh = float(rng.choice([0.7, 1.5, 3.0]))
    x, y = np.meshgrid(np.arange(2 * L), np.arange(2 * L), indexing="ij")
    M = rng.normal(size=(2 * L, 2 * L)) * np.exp(-abs(x - y) / rh)
    H = 1j * (M - M.T)
    chi = int(rng.choice([4, 16, 64])); oc = int(rng.integers(1, L)) if (L > 1 and rng.integers(0, 2)) else None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            CM = pfaffian.correlation_matrix(H, basis="M->M")
            CC = pfaffian.correlation_matrix(H, basis="M->C")
            a = pfaffian.C_to_MPS(CM, {"chi_max": chi}, basis="M", ortho_center=oc, as_tenpy=False)
            b = pfaffian.C_to_MPS(CC, {"chi_max": chi}, basis="C", ortho_center=oc, as_tenpy=False)
        for j in range(L + 1):
            la, lb = np.sort(a.bonds[j].lam), np.sort(b.bonds[j].lam)
            if la.shape != lb.shape or np.abs(la - lb).max() > 1e-9:
                raise AssertionError(f"bond {j}: Schmidt values differ between the two bases")
    except Exception as e:
        bad += 1; print("MISMATCH seed", seed, L, chi, oc, type(e).__name__, str(e)[:120], flush=True)
print("300 cases,", bad, "mismatches")
