"""One case of soak_cells.py with the full traceback.  usage: python tests/soak/cell_case.py <seed>"""
import os, sys, traceback, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_gutzwiller as tg
seed = int(sys.argv[1])
rng = np.random.default_rng(10_000 + seed)
kind = ["ph", "std"][int(rng.integers(0, 2))]; cplx = bool(rng.integers(0, 2)); L = int(rng.choice([2, 4, 6, 8]))
conserve = ["N", "parity"][int(rng.integers(0, 2))]
print(seed, kind, cplx, L, conserve)
try:
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tg.test_infinite_mps_hand_made_cells(kind, seed, cplx, L, conserve)
    print("ok")
except Exception:
    traceback.print_exc(limit=3)
from oracle import gutzwiller_oracle as gw
from temfpy_amd import gutzwiller
rng = np.random.default_rng(seed)
Q = (L // 2) if kind == "std" else (2 * (L // 4) + 2 if conserve == "N" else 0)
cell, q = tg._random_cell(rng, L, Q, conserve, cplx)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    out = gutzwiller.abrikosov_ph(cell) if kind == "ph" else gutzwiller.abrikosov(cell, q_left=1)
M, keep = gw.group_and_project_cell(cell.dense_tensors(), q, Q, kind, conserve, 0, 1)
Bo, So, eta = gw.canonical_form_infinite(M)
for b in range(len(out.lam)):
    a, r = np.sort(out.lam[b])[::-1], np.sort(So[b])[::-1]
    n = min(len(a), len(r))
    print("bond", b, len(a), len(r), "max dev", np.abs(a[:n] - r[:n]).max(), "smallest hip", a[-3:], "oracle", r[-3:])
for j, t in enumerate(out.dense_tensors()):
    X = np.einsum("pab,pcb->ac", t, t.conj())
    d = np.abs(X - np.eye(len(X)))
    print("site", j, "isometry dev", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "lam there", out.lam[j][np.unravel_index(d.argmax(), d.shape)[0]])
print("timings", out.timings)
