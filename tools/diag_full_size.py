import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from tests_inputs import random_hopping
from temfpy_amd import slater
from oracle import slater_oracle as orc
L, chi = 1024, 512
C, N = slater.correlation_matrix(random_hopping(L, 0))
oc = L // 2
mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False)
trunc = orc.as_trunc({"chi_max": chi})
for i in (0, 200, oc - 1, oc, 700, L - 1):
    if i >= oc:
        bra = orc.cut_vectors(C, i + 1, trunc, "R"); ket = orc.cut_vectors(C, i, trunc, "R" if i > oc else "LR")
        site = orc.site_tensor(bra, ket, "right")
    else:
        bra = orc.cut_vectors(C, i, trunc, "L"); ket = orc.cut_vectors(C, i + 1, trunc, "L" if i + 1 < oc else "LR")
        site = orc.site_tensor(bra, ket, "left")
    s = mps.sites[i]
    worst = 0.0; fro = 0.0; tot = 0.0
    for q, r0, r1, c0, c1, blk in s.blocks:
        ref = site.blocks[q][4]
        d = np.abs(np.abs(blk) - np.abs(ref))
        worst = max(worst, d.max() / max(1.0, np.abs(ref).max())); fro += (d**2).sum(); tot += (np.abs(ref)**2).sum()
    near = []
    for c in (bra, ket):
        x = c.x
        ev = np.linalg.eigvalsh(C[:x, :x]) if 0 < x <= oc else (np.linalg.eigvalsh(C[x:, x:]) if x < L else np.zeros(0))
        ev = np.clip(np.minimum(ev, 1 - ev), 1e-300, None)
        near.append(float(np.min(np.abs(np.log10(ev / 1e-12)))) if len(ev) else np.inf)
    print(f"site {i}: max elementwise dev/max {worst:.2e}  rel Frobenius {np.sqrt(fro/tot):.2e}  closest eigenvalue to the cutoff (decades): bra {near[0]:.3f} ket {near[1]:.3f}")
