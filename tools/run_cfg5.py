"""BASELINE config 5: L=512 uniform chain, spinful "PH" Slater -> MPS (1024 sites, chi_max=512), then the
Gutzwiller projection abrikosov_ph to a 512-site spin-1/2 chain (src/examples/gutzwiller.py:15-23 scaled up).
Prints per-stage timings and size-independent checks of the result (development aid / DESIGN numbers)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import uniform_chain  # noqa: E402
from temfpy_amd import slater, gutzwiller  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=512)
ap.add_argument("--chi", type=int, default=512)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--checks", action="store_true")
ap.add_argument("--method", default="sequential")
a = ap.parse_args()
C, N = slater.correlation_matrix(uniform_chain(a.L))
for r in range(a.reps):
    t0 = time.perf_counter()
    mps = slater.C_to_MPS(C, {"chi_max": a.chi}, as_tenpy=False, spinful="PH")
    t1 = time.perf_counter()
    res = gutzwiller.abrikosov_ph(mps, method=a.method)
    t2 = time.perf_counter()
    print(f"rep {r}: Slater->MPS {1e3*(t1-t0):.1f} ms ({mps.L} sites), abrikosov_ph {1e3*(t2-t1):.1f} ms "
          f"({res.L} spins, {res.L/(t2-t1):.1f} sites/s), norm {res.norm:.6e}, max chi {max(res.chi)}, "
          f"S(centre) {res.entanglement_entropy(True)[res.L//2]:.9f}", flush=True)
    print("    " + ", ".join(f"{k} {1e3*v:.1f} ms" for k, v in res.timings.items()), flush=True)
if a.checks:
    worst = 0.0
    for t in res.dense_tensors():
        X = np.einsum("pab,pcb->ac", t, t.conj())
        worst = max(worst, np.abs(X - np.eye(len(X))).max())
    print(f"right-canonical isometry: max deviation {worst:.2e}")
    Bs = res.dense_tensors()
    E = np.ones((1, 1))
    for t in Bs:
        E = np.tensordot(np.tensordot(E, t.conj(), axes=(0, 1)), t, axes=([0, 1], [1, 0]))
    print(f"<psi|psi> = {E[0,0]:.12f}; charges at the ends {res.charges[0].tolist()} {res.charges[-1].tolist()}")
    # S^z symmetry of the half-filled uniform chain: spectrum of bond b is symmetric under q -> -q
    b = res.L // 2
    q, lam = res.charges[b], res.lam[b]
    dev = max(abs(np.sort(lam[q == c])[::-1][: min((q == c).sum(), (q == -c).sum())]
                  - np.sort(lam[q == -c])[::-1][: min((q == c).sum(), (q == -c).sum())]).max() for c in np.unique(q) if c > 0)
    print(f"centre bond: sectors {np.unique(q).tolist()}, sizes {[int((q==c).sum()) for c in np.unique(q)]}, "
          f"S^z -> -S^z asymmetry of the Schmidt values {dev:.2e}")
