"""Dumps the per-bond data of a conversion (entangled eigenvalues, occupation patterns, Schmidt values) into
gpurun_out/ for comparison with the reference in the build container.  usage: dump_bonds.py <case> [<case> ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping, uniform_chain  # noqa: E402
from temfpy_amd import slater  # noqa: E402

CASES = {"cfg3": (lambda: random_hopping(1024, 0), dict(chi_max=512), {}),
         "s1": (lambda: random_hopping(1024, 1), dict(chi_max=512), {}),
         "cfg5": (lambda: uniform_chain(512), dict(chi_max=512), dict(spinful="PH"))}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name in sys.argv[1:]:
    H, tr, kw = CASES[name]
    C, _ = slater.correlation_matrix(H())
    mps = slater.C_to_MPS(C, tr, as_tenpy=False, **kw)
    out = {}
    for b in range(mps.L + 1):
        bd = mps.bonds[b]
        out[f"e{b}"], out[f"m{b}"], out[f"l{b}"], out[f"q{b}"] = bd.e, bd.masks, bd.lam_raw, bd.q_left
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"bonds_{name}.npz"), **out)
    print(name, "dumped", mps.L + 1, "bonds")
