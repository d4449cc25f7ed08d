"""Slater determinant -> MPS (the reference's src/examples/slater.py): convert the ground state of a random hopping
Hamiltonian and compare <c^dagger_i c_j> of the MPS with the correlation matrix it was built from."""
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temfpy_amd import setup_logging, slater  # noqa: E402
from temfpy_amd.utils import HT  # noqa: E402
from dense_mps import correlation_function, state_tensors  # noqa: E402

setup_logging(logging.INFO)


def randomH(L, range=3, seed=0):
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    scale = np.exp(-abs(x - y) / range)
    M = rng.normal(size=(2, L, L), scale=scale)
    M = M[0] + 1j * M[1]
    return M + HT(M)


L, chi = 32, 200
H = randomH(L)
mps = slater.H_to_MPS(H, {"chi_max": chi}, as_tenpy=False)
C, _ = slater.correlation_matrix(H)
CdC = correlation_function(state_tensors(mps), "CdC").T
dev = CdC - C
print("bond dimensions:", mps.chi)
print("max |<c^dag c> - C| =", np.max(np.abs(dev)), " Frobenius", np.linalg.norm(dev))
