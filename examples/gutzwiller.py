"""Gutzwiller projection (the reference's src/examples/gutzwiller.py): half-filled hopping chain of spinful fermions,
particle-hole rotated, projected to spin 1/2; prints the entanglement spectrum of the centre bond by 2 S^z."""
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temfpy_amd import gutzwiller, setup_logging, slater  # noqa: E402

setup_logging(logging.INFO)


def hoppingH(L, t=-1):
    M = np.diag(t * np.ones(L - 1), 1)
    return M + M.T


L, chi = 32, 200
mps_ferm = slater.H_to_MPS(hoppingH(L), {"chi_max": chi}, spinful="PH", as_tenpy=False)
mps_spin = gutzwiller.abrikosov_ph(mps_ferm, inplace=False, return_canonical=True)
print("spin chain of", mps_spin.L, "sites, conserved:", mps_spin.conserve, " norm of the projected state:", mps_spin.norm)
bond = mps_spin.L // 2
lam, q = np.asarray(mps_spin.lam[bond]), np.asarray(mps_spin.charges[bond])
for c in np.unique(q):
    s = -2 * np.log(lam[q == c])
    print(f"2 S^z = {int(c):+d}: entanglement energies", np.round(np.sort(s)[:6], 4))
