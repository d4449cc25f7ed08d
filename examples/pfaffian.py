"""Pfaffian (BCS) state -> MPS (the reference's src/examples/pfaffian.py): random Majorana Hamiltonian, normal and anomalous
correlations of the MPS against the Nambu correlation matrix it was built from."""
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temfpy_amd import pfaffian, setup_logging  # noqa: E402
from dense_mps import correlation_function, state_tensors  # noqa: E402

setup_logging(logging.INFO)


def majorana_random_Hamiltonian(L, range=3, seed=0):
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(2 * L), np.arange(2 * L), indexing="ij")
    scale = np.exp(-abs(x - y) / range)
    M = rng.normal(scale=scale)
    return 1j * (M - M.T)


L, chi = 16, 200
H = majorana_random_Hamiltonian(L)
psi = pfaffian.H_to_MPS(H, {"chi_max": chi}, basis="M", as_tenpy=False)
C = pfaffian.correlation_matrix(H, basis="M->C")
T = state_tensors(psi)
CdC = correlation_function(T, "CdC").T
dev = CdC - C[::2, ::2]
print("<c^dag c>: max deviation", np.max(np.abs(dev)), " Frobenius", np.linalg.norm(dev))
CC = correlation_function(T, "CC").T
dev = CC - C[::2, 1::2]
print("<c c>:     max deviation", np.max(np.abs(dev)), " Frobenius", np.linalg.norm(dev))
