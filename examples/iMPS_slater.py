"""Infinite MPS of an SSH chain from two finite chains that differ by one unit cell (the reference's
src/examples/iMPS_slater.py), with the reference's acceptance check: inserting n more cells into the short chain
reproduces the MPS of the longer chain.  The Gutzwiller projection of the infinite cell is shown as well."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temfpy_amd import gutzwiller, slater  # noqa: E402
from dense_mps import overlap, state_tensors  # noqa: E402


def H(L, t1=-1, t2=-1.5):
    M = t1 * np.ones(L - 1)
    M[1::2] = t2
    M = np.diag(M, 1)
    return M + M.T


trunc_par = dict(chi_max=100)
L_short, cell, n_cell = 64, 2, 8
cut = L_short // 2
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    imps, err = slater.H_to_iMPS(H(L_short), H(L_short + cell), trunc_par, cell, cut, offset=0)
    print("Error metric:", err)
    mps_short = slater.H_to_MPS(H(L_short), trunc_par, ortho_center=cut, as_tenpy=False)
    mps_vlong = slater.H_to_MPS(H(L_short + n_cell * cell), trunc_par, as_tenpy=False)
# A .. A (short chain, left of the cut) | Schmidt values | B .. B (n cells) | B .. B (short chain, right of the cut)
short = [np.asarray(t) for t in mps_short.dense_tensors()]
rec = short[:cut] + [np.asarray(t) for t in imps.dense_tensors()] * n_cell + short[cut:]
rec[cut] = rec[cut] * np.asarray(imps.lam[0])[None, :, None]
Tv = state_tensors(mps_vlong)
ov = overlap(Tv, rec) / np.sqrt(overlap(Tv, Tv).real * overlap(rec, rec).real)
print("Reconstruction overlap:", abs(ov))

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cell_ph, _ = slater.H_to_iMPS(H(24), H(26), dict(chi_max=40), 2, 12, spinful="PH")
spin = gutzwiller.abrikosov_ph(cell_ph)
print("projected infinite cell:", spin.L, "spin sites, chi =", spin.chi, " norm per cell:", spin.norm)
