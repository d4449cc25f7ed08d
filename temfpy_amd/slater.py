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
from .views import MPSTensorData, SchmidtModes, SchmidtVectors  # noqa: F401  (names of the reference's public classes)

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
    (1 - sign(H - mu)) / 2 by a GEMM-only Newton-Schulz iteration (``Engine.negative_projector``), with mu = 0 for
    ``N=None`` (all negative-energy orbitals) and mu found by bisection on the level count for a given ``N``
    (``Engine.lowest_projector``; a degenerate Fermi level raises ValueError)."""
    if device is not None and N is None:
        C, _ = _engine(device).negative_projector(H)
        N = int(np.round(np.trace(C).real))
    elif device is not None:
        C, _, _ = _engine(device).lowest_projector(H, int(N))
        N = int(N)
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
    devices: list | None = None,
    as_tenpy: bool | None = None,
):
    """MPS representation of a Slater determinant from its correlation matrix (slater.py:1216-1353).

    Extra keywords (not in the reference): ``device`` - the GPU that converts; ``devices`` - a list of GPUs
    (e.g. ``["cuda:0", ..., "cuda:7"]``) that share the sites of this one chain, one worker process per
    device (:mod:`temfpy_amd.multi_gpu`; the first such call must come before this process uses a GPU
    itself); ``as_tenpy`` - True: return ``tenpy.networks.mps.MPS`` (ImportError without TeNPy), False:
    return :class:`MPSData`, None: TeNPy object if TeNPy is importable."""
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
    if devices is not None and len(devices) > 1:
        from . import multi_gpu

        mps = multi_gpu.pool(devices).convert(C, trunc_par, ortho_center, unit_cell_width)
    else:
        eng = _engine(devices[0] if devices else device)
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


def _maybe_tenpy(res, as_tenpy):
    """``as_tenpy`` False: the package's container; True: the TeNPy object (ImportError without TeNPy); None: the TeNPy
    object if TeNPy is importable."""
    if as_tenpy is False:
        return res
    try:
        return res.to_tenpy()
    except ImportError:
        if as_tenpy:
            raise
        return res


def H_to_MPS(
    H: np.ndarray,
    trunc_par: dict | StoppingCondition,
    *,
    diag_tol: float = _DIAG_TOL,
    ortho_center: int = None,
    spinful: Literal["simple", "PH", None] = None,
    unit_cell_width: int | None = None,
    device: str = "cuda:0",
    devices: list | None = None,
    as_tenpy: bool | None = None,
):
    """MPS representation of a Slater determinant from its Hamiltonian (slater.py:1568-1627)."""
    C, _ = correlation_matrix(H)
    return C_to_MPS(C, trunc_par, diag_tol=diag_tol, ortho_center=ortho_center, spinful=spinful,
                    unit_cell_width=unit_cell_width, device=device, devices=devices, as_tenpy=as_tenpy)


def C_to_iMPS(
    C_short: np.ndarray,
    C_long: np.ndarray,
    trunc_par: dict | StoppingCondition,
    sites_per_cell: int,
    cut: int,
    *,
    diag_tol: float = _DIAG_TOL,
    unitary_tol: float = 1e-6,
    schmidt_tol: float = 1e-6,
    spinful: Literal["simple", "PH", None] = None,
    offset: int | Literal["auto"] = "auto",
    unit_cell_width: int | None = None,
    device: str = "cuda:0",
    as_tenpy: bool | None = None,
):
    """iMPS representation of a Slater determinant from the correlation matrices of two chains that differ by
    one unit cell (slater.py:1356-1565): same arguments, defaults, offset rules and exceptions.

    As in the reference the last tensor of the unit cell is expressed in the right Schmidt vectors of the SHORT chain
    (slater.py:1508-1518) and no right-hand errors are reported (slater.py:1563); the first tensor carries the Procrustes
    rotation of the left Schmidt-vector overlaps (slater.py:1538-1553).  Difference in method, stated rather than hidden:
    the reference gets those overlaps from determinant formulas without environment tensors (slater.py:1443-1446,
    1023-1024); here both chains are converted in full (30 ms each at L = 1024) with their orthogonality centre at ``cut``
    and the overlaps come from the transfer matrices of the two MPS (:func:`temfpy_amd.iMPS.MPS_to_iMPS` with
    ``right="project"``), i.e. they are overlaps of the truncated Schmidt vectors.  Same state up to the truncation
    (acceptance check of src/examples/iMPS.py:27-38 in tests/test_gpu_imps.py)."""
    from . import iMPS

    trunc_par = to_stopping_condition(trunc_par)
    if unit_cell_width is None:                                # verified *before* doubling C (slater.py:1449-1453)
        unit_cell_width = sites_per_cell
    elif sites_per_cell % unit_cell_width != 0:
        raise ValueError(f"{unit_cell_width = } does not divide {sites_per_cell = }")
    if spinful == "simple":
        if offset == "auto":                                   # slater.py:1456-1461
            offset = 2 * round(np.trace(C_short[:cut, :cut]).real)
        else:
            offset *= 2
    elif spinful not in ("PH", None):
        raise ValueError(f"`spinful` must be 'simple', 'PH', or `None`, got {spinful!r}")
    mult = 1 if spinful is None else 2
    L_short, L_long = mult * len(C_short), mult * len(C_long)
    assert np.shape(C_short) == (len(C_short),) * 2, f"Got non-square {np.shape(C_short)} correlation matrix"
    assert np.shape(C_long) == (len(C_long),) * 2, f"Got non-square {np.shape(C_long)} correlation matrix"
    assert L_short + mult * sites_per_cell == L_long, (
        "The given two MPS must differ by one unit cell, got "
        f"{L_long} - {L_short} != {mult * sites_per_cell}")
    if offset == "auto":                                       # slater.py:1491 (after doubling)
        C2 = C_short if spinful is None else spinful_correlation_matrix(C_short, spinful == "PH")
        offset = round(np.trace(C2[: mult * cut, : mult * cut]).real)
    mps_s = C_to_MPS(C_short, trunc_par, diag_tol=diag_tol, ortho_center=mult * cut, spinful=spinful, device=device,
                     as_tenpy=False)
    mps_l = C_to_MPS(C_long, trunc_par, diag_tol=diag_tol, ortho_center=mult * cut, spinful=spinful, device=device,
                     as_tenpy=False)
    res, err = iMPS.MPS_to_iMPS(mps_s, mps_l, mult * sites_per_cell, mult * cut, unitary_tol=unitary_tol,
                                schmidt_tol=schmidt_tol, offset=offset, unit_cell_width=mult * sites_per_cell,
                                device=device, right="project")
    res.unit_cell_width = unit_cell_width
    return _maybe_tenpy(res, as_tenpy), err


def H_to_iMPS(
    H_short: np.ndarray,
    H_long: np.ndarray,
    trunc_par: dict | StoppingCondition,
    sites_per_cell: int,
    cut: int,
    *,
    diag_tol: float = _DIAG_TOL,
    unitary_tol: float = 1e-6,
    schmidt_tol: float = 1e-6,
    spinful: Literal["simple", "PH", None] = None,
    offset: int | Literal["auto"] = "auto",
    unit_cell_width: int | None = None,
    device: str = "cuda:0",
    as_tenpy: bool | None = None,
):
    """iMPS representation of a Slater determinant from single-particle Hamiltonians (slater.py:1630-1734)."""
    C_short, _ = correlation_matrix(H_short)
    C_long, _ = correlation_matrix(H_long)
    return C_to_iMPS(C_short, C_long, trunc_par, sites_per_cell, cut, diag_tol=diag_tol, unitary_tol=unitary_tol,
                     schmidt_tol=schmidt_tol, spinful=spinful, offset=offset, unit_cell_width=unit_cell_width,
                     device=device, as_tenpy=as_tenpy)


__all__ = ["C_to_MPS", "H_to_MPS", "C_to_iMPS", "H_to_iMPS", "correlation_matrix", "spinful_correlation_matrix", "MPSData"]
