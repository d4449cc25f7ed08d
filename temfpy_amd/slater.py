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

    As in the reference every tensor is a determinant formula between the Schmidt vectors of two cuts and no chain is converted
    in full (slater.py:1443-1446): ONE sweep decomposes the cuts ``cut .. cut + sites_per_cell - 1`` of the long chain and the
    cut of the short chain, the last tensor of the unit cell takes the right Schmidt vectors of the SHORT chain as its bra
    (slater.py:1508-1518), the first one carries the Procrustes rotation of the overlaps of the left Schmidt vectors
    (slater.py:1538-1553; overlaps without a physical leg, :1023-1024), and no right-hand errors are reported (:1563).
    (``TMF_IMPS=transfer`` keeps the earlier method for A/B: both chains converted in full, overlaps from their transfer
    matrices, :func:`temfpy_amd.iMPS.MPS_to_iMPS` with ``right="project"`` - the same state up to the truncation.)

    The short chain enters the sweep as ``diag(0, C_short)`` of the long chain's size: ``sites_per_cell`` empty, decoupled
    sites in front leave its Schmidt decomposition at the cut unchanged and line its right block up with the one of the long
    chain, so that both chains share one batch of cut problems.

    The reference's example inserts the unit cell into a SEPARATELY converted short chain (src/examples/iMPS.py:27-38), which
    works there because LAPACK returns the same eigenvectors for the same block twice.  Two different sweeps do not share
    rounding, so what rounding decides is fixed by rule in every sweep: the phases of the entangled orbitals and the basis
    inside groups of eigenvalues C does not tell apart (``tmf_canonical_gauge_batched``), and the order of occupation
    patterns of equal weight (``tmf_cut_vectors``).  ``C_to_MPS(C_short, ortho_center=cut)`` and this function then number
    and sign the Schmidt vectors of the cut alike, also for two identical spin species (tests/test_gpu_imps.py)."""
    import os

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
    C2s = np.asarray(C_short) if spinful is None else spinful_correlation_matrix(np.asarray(C_short), spinful == "PH")
    if offset == "auto":                                       # slater.py:1491 (after doubling)
        offset = round(np.trace(C2s[: mult * cut, : mult * cut]).real)
    if os.environ.get("TMF_IMPS", "determinants") == "transfer":
        mps_s = C_to_MPS(C_short, trunc_par, diag_tol=diag_tol, ortho_center=mult * cut, spinful=spinful, device=device,
                         as_tenpy=False)
        mps_l = C_to_MPS(C_long, trunc_par, diag_tol=diag_tol, ortho_center=mult * cut, spinful=spinful, device=device,
                         as_tenpy=False)
        res, err = iMPS.MPS_to_iMPS(mps_s, mps_l, mult * sites_per_cell, mult * cut, unitary_tol=unitary_tol,
                                    schmidt_tol=schmidt_tol, offset=offset, unit_cell_width=mult * sites_per_cell,
                                    device=device, right="project")
        res.unit_cell_width = unit_cell_width
        return _maybe_tenpy(res, as_tenpy), err
    from .engine import _drive

    C2l = np.asarray(C_long) if spinful is None else spinful_correlation_matrix(np.asarray(C_long), spinful == "PH")
    spc, x = mult * sites_per_cell, mult * cut
    assert 0 < x < L_short, f"Invalid entanglement cut {x}"
    dt = np.result_type(C2s.dtype, C2l.dtype, np.float64)
    emb = np.zeros((L_long, L_long), dt)
    emb[spc:, spc:] = C2s
    cross = [dict(mode=1, phys=True, bra=(1, x + spc), ket=(0, x + spc - 1)),         # last tensor (slater.py:1513)
             dict(mode=0, phys=False, bra=(1, x + spc), ket=(0, x), skip=spc)]         # gauge overlaps (slater.py:1538)
    eng = _engine(device)
    eng.checks = testing.TEST_ACTION != "pass"
    res = _drive(eng.run_gen(np.ascontiguousarray(C2l, dt), trunc_par, x, L_long, site_range=(x, x + spc - 1),
                             second=dict(C=emb, oc=x + spc, cross=cross)))
    testing.report_schmidt_checks(res.info["checks"], diag_tol)     # both reference cuts (slater.py:419-420 via :1499-1505)
    cell, err = iMPS.cell_from_determinants(res.shards[0], L_long, x, spc, int(offset), unit_cell_width,
                                            unitary_tol=unitary_tol, schmidt_tol=schmidt_tol, device=device)
    return _maybe_tenpy(cell, as_tenpy), err


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
