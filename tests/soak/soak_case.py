"""One case of tests/soak/soak_small.py in detail.  usage: python tests/soak/soak_case.py <seed>"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import slater_oracle as orc
from temfpy_amd import slater
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
max_L = int(sys.argv[2]) if len(sys.argv) > 2 else 28
L = int(rng.integers(2, max_L + 1)); rng_h = float(rng.choice([0.7, 1.5, 3.0, 6.0])); cplx = bool(rng.integers(0, 2))
x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
H = M[0] + (1j * M[1] if cplx else 0); H = H + H.conj().T
N = int(rng.integers(0, L + 1)) if rng.integers(0, 3) == 0 else None
spinful = [None, None, None, None, "simple", "PH"][int(rng.integers(0, 6))]
chi = int(rng.choice([2, 5, 16, 40, 128] if max_L <= 28 else [16, 40, 128, 300]))
Lf = L * (1 if spinful is None else 2)
oc = int(rng.integers(1, Lf)) if (Lf > 1 and rng.integers(0, 2)) else None
trunc = {"chi_max": chi}
if os.environ.get("SOAK_TRUNC"):
    trunc["svd_min"] = float(rng.choice([1e-3, 1e-5, 1e-6, 1e-7]))
    trunc["degeneracy_tol"] = float(rng.choice([1e-12, 1e-9, 1e-6]))
print(f"seed {seed}: L={L} range={rng_h} complex={cplx} N={N} spinful={spinful} trunc={trunc} oc={oc}")
C, Np = orc.correlation_matrix(H, N)
print("particles", Np, "eigenvalues of H closest to 0:", np.sort(np.abs(np.linalg.eigvalsh(H)))[:3])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cuts, sites = orc.c_to_mps(C, dict(trunc), ortho_center=oc, spinful=spinful)
    try:
        mps = slater.C_to_MPS(C, dict(trunc), ortho_center=oc, spinful=spinful, as_tenpy=False)
    except Exception as e:
        print("HIP path raised", type(e).__name__, e)
        for b, c_ in enumerate(cuts):
            print(b, "oracle k", len(c_.e), "e", np.array2string(np.asarray(c_.e), precision=3, max_line_width=200))
        sys.exit(0)
for b in range(Lf + 1):
    a, o = mps.bonds[b], cuts[b]
    same = np.array_equal(a.sets, o.sets)
    ea, eo = np.asarray(a.e), np.asarray(o.e)
    print(b, "k", len(ea), len(eo), "chi", len(a.lam), len(o.lam), "sets equal", same,
          "max|de|", (np.abs(ea - eo).max() if len(ea) == len(eo) and len(ea) else None))
    if len(ea) != len(eo):
        w = lambda e: np.minimum(e, 1 - e)
        print("   weakest hip   ", np.sort(w(ea))[:16])
        print("   weakest oracle", np.sort(w(eo))[:16])
    if not same and len(ea) == len(eo):
        print("   e hip   ", np.array2string(ea, precision=15, max_line_width=250))
        print("   e oracle", np.array2string(eo, precision=15, max_line_width=250))
eng = slater._engine("cuda:0")
print("range finder: width", eng.range_width, "iterations", eng.range_iterations_used, "smallest captured sigma", eng.range_floor)
occ = oc or Lf // 2
T1, T2 = orc.dense_tensors(cuts, sites), mps.dense_tensors()
n1 = orc.mps_overlap(T1, cuts[occ].lam, T1, cuts[occ].lam, occ); n2 = orc.mps_overlap(T2, mps.lam[occ], T2, mps.lam[occ], occ)
print("norms", abs(n1), abs(n2), "1 - overlap", 1 - abs(orc.mps_overlap(T1, cuts[occ].lam, T2, mps.lam[occ], occ)) / np.sqrt(abs(n1 * n2)))
for i, (a, b) in enumerate(zip(T1, T2)):
    print(" site", i, "max | |a| - |b| |", np.abs(np.abs(a) - np.abs(b)).max())
