"""Synthetic inputs of the BASELINE configs (seeded restatements of src/examples/*.py)."""
import numpy as np


def random_hopping(L, seed, rng_range=3.0):
    """src/examples/slater.py:15-20 with a seeded generator."""
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_range)
    M = M[0] + 1j * M[1]
    return M + M.T.conj()


def uniform_chain(L, t=-1.0):
    """src/examples/gutzwiller.py:10-12."""
    M = np.diag(t * np.ones(L - 1), 1)
    return M + M.T
