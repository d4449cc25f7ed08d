"""Slater determinant -> MPS with TeMFpy's entry points (temfpy/slater.py), on MI355X.

``C_to_MPS`` / ``H_to_MPS`` keep the reference's signatures, defaults and exceptions
(slater.py:1216-1224, :1568-1576).  They return a ``tenpy.networks.mps.MPS`` when TeNPy is
importable and ``as_tenpy`` is not False, else a :class:`temfpy_amd.mps_data.MPSData` with the
same tensors, Schmidt values, charges and canonical form.
"""
from __future__ import annotations

import logging
from typing import Literal

import numpy as np

from .schmidt_utils import StoppingCondition, to_stopping_condition
from . import testing
from .testing import _DIAG_TOL
from .utils import HT
from .mps_data import MPSData

logger = logging.getLogger(__name__)

_ENGINES = {}


def _engine(device):
    from .engine import Engine

    if device not in _ENGINES:
        _ENGINES[device] = Engine(device)
    return _ENGINES[device]


def correlation_matrix(H: np.ndarray, N: int | None = None, *, device: str | None = None) -> tuple[np.ndarray, int]:
    """Ground-state correlation matrix of a mean-field Hamiltonian (slater.py:1150-1180).

    Outside the timed C -> MPS path.  Default: host LAPACK like the reference.
    ``device="cuda:0"`` (extra keyword): the occupied-orbital projector is computed on the GPU as
    (1 - sign(H)) / 2 by a GEMM-only Newton-Schulz iteration (``Engine.negative_projector``); only for
    ``N=None`` (occupy all negative-energy orbitals), where no eigenvalue ordering is needed."""
    if device is not None and N is None:
        C, _ = _engine(device).negative_projector(H)
        N = int(np.round(np.trace(C).real))
    else:
        e, v = np.linalg.eigh(H)
        if N is None:
            occupied = e < 0
            v = v[:, occupied]
            N = int(occupied.sum())
        else:
            v = v[:, :N]
        C = v @ HT(v)
    if np.iscomplexobj(C) and np.allclose(C.imag, 0.0, rtol=0, atol=1e-14):
        C = C.real
    return C, N


def spinful_correlation_matrix(C: np.ndarray, ph: bool = True):
    """Enlarged correlation matrix for spinful fermions (slater.py:1183-1213)."""
    n, m = C.shape
    assert n == m, f"Got non-square {C.shape} correlation matrix"
    C2 = np.zeros((2 * n, 2 * n), dtype=C.dtype)
    C2[::2, ::2] = C
    C2[1::2, 1::2] = (np.eye(n) - C) if ph else C
    return C2


def C_to_MPS(
    C: np.ndarray,
    trunc_par: dict | StoppingCondition,
    *,
    diag_tol: float = _DIAG_TOL,
    ortho_center: int = None,
    spinful: Literal["simple", "PH", None] = None,
    unit_cell_width: int | None = None,
    device: str = "cuda:0",
    as_tenpy: bool | None = None,
):
    """MPS representation of a Slater determinant from its correlation matrix (slater.py:1216-1353)."""
    trunc_par = to_stopping_condition(trunc_par)
    if unit_cell_width is None:
        unit_cell_width = len(C)
    elif len(C) % unit_cell_width != 0:
        raise ValueError(f"{unit_cell_width = } does not divide system size {len(C)}")  # slater.py:1272
    if spinful == "simple":
        C = spinful_correlation_matrix(C, False)
    elif spinful == "PH":
        C = spinful_correlation_matrix(C, True)
    elif spinful is not None:
        raise ValueError(f"`spinful` must be 'simple', 'PH', or `None`, got {spinful!r}")
    C = np.asarray(C)
    L = len(C)
    assert C.shape == (L, L), f"Got non-square {C.shape} correlation matrix"
    ortho_center = ortho_center or L // 2  # slater.py:1291
    logger.info("Central bond %d", ortho_center)
    eng = _engine(device)
    eng.checks = testing.TEST_ACTION != "pass"  # testing.py:146-147
    mps = eng.run(C, trunc_par, ortho_center, unit_cell_width)
    testing.report_schmidt_checks(mps.info["checks"], diag_tol)  # slater.py:419-420
    if as_tenpy is False:
        return mps
    try:
        return mps.to_tenpy()
    except ImportError:
        if as_tenpy:
            raise
        return mps


def H_to_MPS(
    H: np.ndarray,
    trunc_par: dict | StoppingCondition,
    *,
    diag_tol: float = _DIAG_TOL,
    ortho_center: int = None,
    spinful: Literal["simple", "PH", None] = None,
    unit_cell_width: int | None = None,
    device: str = "cuda:0",
    as_tenpy: bool | None = None,
):
    """MPS representation of a Slater determinant from its Hamiltonian (slater.py:1568-1627)."""
    C, _ = correlation_matrix(H)
    return C_to_MPS(C, trunc_par, diag_tol=diag_tol, ortho_center=ortho_center, spinful=spinful,
                    unit_cell_width=unit_cell_width, device=device, as_tenpy=as_tenpy)


__all__ = ["C_to_MPS", "H_to_MPS", "correlation_matrix", "spinful_correlation_matrix", "MPSData"]
