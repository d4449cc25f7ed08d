"""Small utilities with the reference's names (temfpy/utils.py)."""
import logging

import numpy as np


def HT(M: np.ndarray) -> np.ndarray:
    """Hermitian conjugate (utils.py:8-10)."""
    return M.T.conj()


def n_slice(x: slice) -> int:
    """Number of elements of a slice (utils.py:13-16)."""
    return (x.stop - x.start) // (x.step or 1)


def normalize_SV(lam: np.ndarray, logger: logging.Logger) -> np.ndarray:
    """utils.py:99-103."""
    norm = np.linalg.norm(lam)
    logger.info(f"Norm of Schmidt values: {norm}")
    return lam / norm
