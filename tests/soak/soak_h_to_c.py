"""Randomised soak of slater.correlation_matrix(H, N, device=...) (bisection on the chemical potential) against host eigh:
random sizes, ranges, fillings, some Hamiltonians with exactly degenerate levels.  Development aid."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from temfpy_amd import slater
bad = deg = 0
for seed in range(300):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 121)); cplx = bool(rng.integers(0, 2))
    M = rng.normal(size=(2, L, L)) * np.exp(-abs(np.subtract.outer(np.arange(L), np.arange(L))) / float(rng.choice([0.7, 3.0, 50.0])))
    H = M[0] + (1j * M[1] if cplx else 0); H = H + H.conj().T
    if rng.integers(0, 4) == 0:
        H = np.round(H * 2) / 2          # many exactly degenerate levels
    N = int(rng.integers(0, L + 1))
    e = np.linalg.eigvalsh(H)
    gap = (e[N] - e[N - 1]) if 0 < N < L else 1.0
    try:
        C1, N1 = slater.correlation_matrix(H, N, device="cuda:0")
        C0, N0 = slater.correlation_matrix(H, N)
        if gap < 1e-9 * max(abs(e).max(), 1e-300):
            print("seed", seed, "degenerate Fermi level accepted: gap", gap); bad += 1; continue
        d = np.abs(C1 - C0).max()
        if d > 1e-9 / max(gap / max(abs(e).max(), 1e-300), 1e-6) * 1e-3 + 1e-10:
            print("seed", seed, "L", L, "N", N, "gap", gap, "dev", d); bad += 1
    except ValueError as ex:
        if gap < 1e-6 * max(abs(e).max(), 1e-300) or "zero Hamiltonian" in str(ex):
            deg += 1
        else:
            print("seed", seed, "L", L, "N", N, "gap", gap, "raised", str(ex)[:100]); bad += 1
    except Exception as ex:
        print("seed", seed, "L", L, "N", N, "gap", gap, type(ex).__name__, str(ex)[:120]); bad += 1
print("300 cases,", bad, "mismatches,", deg, "degenerate Fermi levels refused")
