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
ap.add_argument("--method", default="parallel")
ap.add_argument("--cpu-sample", type=int, default=0,
                help="also time the CPU oracle (charge-block restatement of TeNPy's canonical_form_finite) on a chain of "
                     "this many fermion sites per species at the same chi_max: the per-site cost is size independent once chi saturates")
ap.add_argument("--json", default=None, help="write the measurements to this file")
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
out = {"workload": f"L={a.L} uniform chain, spinful PH, chi_max={a.chi} -> abrikosov_ph ({a.method})",
       "slater_ms": round(1e3 * (t1 - t0), 1), "gutzwiller_ms": round(1e3 * (t2 - t1), 1), "spins": res.L,
       "spin_sites_per_s": round(res.L / (t2 - t1), 1), "norm": res.norm, "max_chi": max(res.chi),
       "S_centre": float(res.entanglement_entropy(True)[res.L // 2]),
       "stage_ms": {k: round(1e3 * v, 1) for k, v in res.timings.items()}}
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

if a.cpu_sample:
    from oracle import gutzwiller_oracle as gw
    Cs, _ = slater.correlation_matrix(uniform_chain(a.cpu_sample))
    m2 = slater.C_to_MPS(Cs, {"chi_max": a.chi}, as_tenpy=False, spinful="PH")
    T = m2.dense_tensors()
    q = [np.asarray(b.q_left) for b in m2.bonds]
    t0 = time.perf_counter()
    M, keep = gw.group_and_project(T, q, m2.lam[m2.ortho_center], m2.ortho_center, "ph")
    ql = gw.spin_charges(q, keep)
    B, S, Q, nrm = gw.canonical_form_finite_blocks(M, ql)
    dt = time.perf_counter() - t0
    r2 = gutzwiller.abrikosov_ph(m2, method=a.method)
    dS = max(np.abs(np.sort(x)[::-1][:min(len(x), len(y))] - np.sort(y)[::-1][:min(len(x), len(y))]).max()
             for x, y in zip(r2.lam, S))
    print(f"CPU oracle (NumPy/LAPACK, charge blocks) on {len(M)} spins at chi_max={a.chi}: {dt:.2f} s = {len(M)/dt:.1f} spin sites/s; "
          f"HIP on the same chain: {len(M)/sum(r2.timings.values()):.1f}; max |d lambda| = {dS:.1e}")
    out["cpu_oracle"] = {"spins": len(M), "seconds": round(dt, 2), "spin_sites_per_s": round(len(M) / dt, 1),
                         "max_abs_dlambda_vs_hip": float(dS)}
if a.json:
    import json
    json.dump(out, open(a.json, "w"), indent=1)
