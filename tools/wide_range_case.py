"""A chain whose cuts carry more than 64 entangled orbitals (long-range hopping): the range finder escalates to 128 / 256
columns and the orthogonalisation runs on the general Householder kernel.  Prints widths, time and invariants.
usage: python tools/wide_range_case.py [L] [range] [chi]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from temfpy_amd import slater
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng_h = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
chi = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(0)
x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
H = M[0] + 1j * M[1]; H = H + H.conj().T
C, N = slater.correlation_matrix(H)
eng = slater._engine("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(2):
        t0 = time.perf_counter()
        mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False)
        mps.wait() if hasattr(mps, "wait") else None
        dt = time.perf_counter() - t0
kmax = max(len(b.e) for b in mps.bonds)
print(f"L={L} range={rng_h} chi={chi}: {dt*1e3:.1f} ms, range-finder width {eng.range_width}, iterations {eng.range_iterations_used}, "
      f"most entangled orbitals at a cut {kmax}, max chi {max(mps.chi)}")
lam = [np.asarray(b.lam) for b in mps.bonds]
print("max |sum lam^2 - 1| =", max(abs((l_ ** 2).sum() - 1) for l_ in lam))
S = mps.entanglement_entropy(all_bonds=True)
print("S(centre) =", S[L // 2], " finite:", np.isfinite(S).all())
