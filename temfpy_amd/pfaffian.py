"""BCS / Pfaffian mean-field state -> MPS with TeMFpy's entry points (temfpy/pfaffian.py), on MI355X.

Same names, arguments and defaults as the reference for the converter entry points
(pfaffian.py:1785-1793, :2094-2102, :302-304) and the basis-change helpers (pfaffian.py:75-184).
Returns a ``tenpy.networks.mps.MPS`` when TeNPy is importable and ``as_tenpy`` is not False (assembly
as pfaffian.py:1750-1778, self-checked at run time: :meth:`PfMPSData.to_tenpy`), else a
:class:`temfpy_amd.engine_pf.PfMPSData` (parity-graded blocks, Schmidt values, vacuum parities).
"""
from __future__ import annotations

import logging

import numpy as np

from .schmidt_utils import StoppingCondition, to_stopping_condition
from . import testing
from .testing import _DIAG_TOL, assert_allclose, assert_array_less
from .utils import HT

logger = logging.getLogger(__name__)
_ENGINES = {}

_M_C2M = np.array([[1, 1], [1j, -1j]]) / 2**0.5   # pfaffian.py:93
_M_M2C = np.array([[1, -1j], [1, 1j]]) / 2**0.5   # pfaffian.py:122


def vector_C2M(v: np.ndarray) -> np.ndarray:
    """pfaffian.py:75-101."""
    n = v.shape[0]
    assert n % 2 == 0, "Got vector(s) of odd size (cannot be Nambu)"
    w = v.reshape(n // 2, 2, *v.shape[1:])
    return np.einsum("xa...,ca->xc...", w, _M_C2M).reshape(n, *v.shape[1:])


def vector_M2C(v: np.ndarray) -> np.ndarray:
    """pfaffian.py:104-130."""
    n = v.shape[0]
    assert n % 2 == 0, "Got vector(s) of odd size (cannot be Nambu)"
    w = v.reshape(n // 2, 2, *v.shape[1:])
    return np.einsum("xa...,ca->xc...", w, _M_M2C).reshape(n, *v.shape[1:])


def matrix_C2M(H: np.ndarray) -> np.ndarray:
    """pfaffian.py:133-157."""
    n, m = H.shape
    assert n % 2 == 0 and m % 2 == 0, "Got a matrix with odd side length (cannot be Nambu)"
    return np.einsum("xayb,ca,db->xcyd", H.reshape(n // 2, 2, m // 2, 2), _M_C2M, _M_C2M.conj()).reshape(n, m)


def matrix_M2C(H: np.ndarray) -> np.ndarray:
    """pfaffian.py:160-184."""
    n, m = H.shape
    assert n % 2 == 0 and m % 2 == 0, "Got a matrix with odd side length (cannot be Nambu)"
    return np.einsum("xayb,ca,db->xcyd", H.reshape(n // 2, 2, m // 2, 2), _M_M2C, _M_M2C.conj()).reshape(n, m)


def assert_nambu(C, basis=None, offset=None, name="", rtol=0, atol=1e-10):
    """pfaffian.py:189-286: Hermitise and enforce the Nambu structure (checks follow TEST_ACTION)."""
    n, m = C.shape
    assert n == m > 0, f"Got non-square {name}"
    assert n % 2 == 0, f"Got {name} with odd side length (cannot be Nambu)"
    n //= 2
    tol = dict(atol=atol, rtol=rtol)
    Ct = np.array(C.T, order="C", copy=True)   # HT(C), once and row-major (a COPY also when C.T is row-major already)
    Ct = np.conjugate(Ct, out=Ct) if np.iscomplexobj(Ct) else Ct
    assert_allclose(C, Ct, **tol, err_msg=f"{name} is not Hermitian")
    C = C + Ct
    C *= 0.5
    if basis == "M":
        real = np.eye(2 * n) * offset / 2
        assert_allclose(C.real, real, **tol, err_msg="Unexpected real parts in Majorana basis")
        if np.iscomplexobj(C):
            C.real[...] = real        # (in place: C is this function's own array)
        else:
            C = real + 1j * C.imag
    elif basis == "C":
        err = f"{name.capitalize()} is not Nambu symmetric"
        assert_allclose(C[::2, ::2], offset * np.eye(n) - C[1::2, 1::2].conj(), **tol, err_msg=err)
        assert_allclose(C[1::2, ::2], -C[::2, 1::2].conj(), **tol, err_msg=err)
        if np.allclose(C.imag, 0, **tol):
            C = C.real
    elif basis is not None:
        raise ValueError("Invalid `basis` " + repr(basis))
    return C


_ENGINES_HC = {}


def correlation_matrix(H: np.ndarray, basis: str | None = None, *, rtol: float = 0, atol: float = 1e-10,
                       device: str | None = None):
    """Ground-state Nambu correlation matrix of a BdG Hamiltonian (pfaffian.py:302-393).
    Outside the timed C -> MPS path; host LAPACK like the reference, or with ``device="cuda:0"`` (extra
    keyword) the negative-energy projector from the GEMM-only sign iteration of
    ``Engine.negative_projector`` (zero modes make it fail to converge -> RuntimeError, like the
    reference's check at pfaffian.py:372-377)."""
    assert basis in [None, "M->M", "M->C", "C->M", "C->C"], f"Invalid basis spec {basis!r}, should be of form '[MC]->[MC]'"
    tol = dict(rtol=rtol, atol=atol)
    H = assert_nambu(H, None if basis is None else basis[0], offset=0, name="Hamiltonian", **tol)
    n = len(H) // 2
    if device is not None:
        from .engine import Engine

        eng = _ENGINES_HC.setdefault(device, Engine(device))
        C, _ = eng.negative_projector(np.asarray(H, complex))
        if basis == "C->M":
            C = matrix_C2M(C)
        elif basis == "M->C":
            C = matrix_M2C(C)
        return assert_nambu(C, None if basis is None else basis[3], offset=1, name="correlation matrix", **tol)
    e, v = np.linalg.eigh(H)
    assert_allclose(e + e[::-1], 0, **tol)
    if np.any(abs(e) < atol):
        raise RuntimeError("Some energy eigenvalues are zero. You need to construct\nyour own correlation matrix!\n"
                           f"Middle 10 eigenvalues:\n{e[n - 5: n + 5, None]}")
    assert_array_less(e[:n], 0, "Lower half of eigenvalues is not all negative")
    v = v[:, :n]
    if basis == "C->M":
        v = vector_C2M(v)
    elif basis == "M->C":
        v = vector_M2C(v)
    C = v @ HT(v)
    return assert_nambu(C, None if basis is None else basis[3], offset=1, name="correlation matrix", **tol)


def parity(V: np.ndarray, *, tol: float = 1e-12, device: str = "cuda:0") -> int:
    """Fermion parity of a Bogoliubov vacuum from the singular values of ``V`` (pfaffian.py:396-456): the
    values strictly between 0 and 1 come in pairs, so the parity of the number of singular values above the
    largest gap is that of the completely filled modes.  The singular values come from the GPU
    (``utils._device_svd``)."""
    from .utils import _device_svd

    V = np.asarray(V)
    if len(V) == 0:
        return 0
    if len(V) == 1:
        v = V.item()
        if np.isclose(v, 0.0, rtol=0, atol=tol):
            return 0
        if np.isclose(abs(v), 1.0, rtol=0, atol=tol):
            return 1
        raise RuntimeError("Invalid 1x1 V")
    (s,) = _device_svd([V], device, False)
    if len(V) > 2:
        n = np.argmax(-np.diff(s))
        return int((n + 1) % 2)
    if np.allclose(s, [1.0, 0.0], rtol=0, atol=tol):
        return 1
    if np.isclose(s[0], s[1], rtol=0, atol=tol):
        return 0
    raise ValueError("Invalid 2x2 V")


def C_to_MPS(C: np.ndarray, trunc_par: dict | StoppingCondition, *, basis: str, diag_tol: float = _DIAG_TOL,
             ortho_center: int = None, unit_cell_width: int | None = None, device: str = "cuda:0",
             as_tenpy: bool | None = None):
    """MPS of a BCS / Pfaffian state from its Nambu correlation matrix (pfaffian.py:1785-1921)."""
    from .engine_pf import PfEngine

    trunc_par = to_stopping_condition(trunc_par)
    C = np.asarray(C)
    L = len(C) // 2
    if unit_cell_width is None:
        unit_cell_width = L
    elif L % unit_cell_width != 0:
        raise ValueError(f"{unit_cell_width = } does not divide system size {L}")  # pfaffian.py:1839
    if basis == "C":
        C = matrix_C2M(C)  # pfaffian.py:750-751
    elif basis != "M":
        raise ValueError(f"Argument `basis` must be 'M' or 'C', got {basis!r}")
    C = assert_nambu(np.asarray(C, complex), "M", offset=1, name="correlation matrix", atol=trunc_par.svd_min**2)
    ortho_center = ortho_center or L // 2
    logger.info("Central bond %d", ortho_center)
    if device not in _ENGINES:
        _ENGINES[device] = PfEngine(device)
    eng = _ENGINES[device]
    eng.checks = testing.TEST_ACTION != "pass"
    mps = eng.run(C, trunc_par, ortho_center, unit_cell_width)
    testing.report_schmidt_checks(mps.info["checks"], diag_tol)  # pfaffian.py:919
    # as_tenpy as in slater.C_to_MPS: True -> tenpy.networks.mps.MPS (ImportError without TeNPy), False -> PfMPSData,
    # None -> the TeNPy object if TeNPy is importable (pfaffian.py:1916-1919)
    if as_tenpy is False:
        return mps
    try:
        return mps.to_tenpy()
    except ImportError:
        if as_tenpy:
            raise
        return mps


def H_to_MPS(H: np.ndarray, trunc_par: dict | StoppingCondition, *, basis: str, diag_tol: float = _DIAG_TOL,
             ortho_center: int = None, unit_cell_width: int | None = None, device: str = "cuda:0",
             as_tenpy: bool | None = None):
    """pfaffian.py:2094-2148."""
    C = correlation_matrix(H, f"{basis}->M")
    return C_to_MPS(C, trunc_par, basis="M", diag_tol=diag_tol, ortho_center=ortho_center,
                    unit_cell_width=unit_cell_width, device=device, as_tenpy=as_tenpy)


def C_to_iMPS(C_short: np.ndarray, C_long: np.ndarray, trunc_par: dict | StoppingCondition, sites_per_cell: int,
              cut: int, *, basis: str, diag_tol: float = _DIAG_TOL, unitary_tol: float = 1e-6,
              schmidt_tol: float = 1e-6, unit_cell_width: int | None = None, device: str = "cuda:0",
              as_tenpy: bool | None = None):
    """iMPS representation of a Nambu mean-field state from the correlation matrices of two chains that differ by
    one unit cell (pfaffian.py:1924-2091): same arguments, defaults and exceptions.  As in the reference the last
    tensor of the unit cell is expressed in the right Schmidt vectors of the SHORT chain (pfaffian.py:2039-2056) and no
    right-hand errors are reported (pfaffian.py:2090).  Difference in method, stated rather than hidden (as in
    ``temfpy_amd.slater.C_to_iMPS``): both chains are converted in full with their orthogonality centre at ``cut`` and the
    Schmidt-vector overlaps come from the transfer matrices of the two MPS (:func:`temfpy_amd.iMPS.MPS_to_iMPS` with
    ``right="project"``), whereas the reference uses its Pfaffian overlap formulas without environment tensors."""
    from . import iMPS

    trunc_par = to_stopping_condition(trunc_par)
    if unit_cell_width is None:
        unit_cell_width = sites_per_cell
    elif sites_per_cell % unit_cell_width != 0:
        raise ValueError(f"{unit_cell_width = } does not divide {sites_per_cell = }")
    L_short, L_long = len(C_short) // 2, len(C_long) // 2
    assert L_short + sites_per_cell == L_long, (
        "The given two MPS must differ by one unit cell, got " f"{L_long} - {L_short} != {sites_per_cell}")
    mps_s = C_to_MPS(C_short, trunc_par, basis=basis, diag_tol=diag_tol, ortho_center=cut, device=device, as_tenpy=False)
    mps_l = C_to_MPS(C_long, trunc_par, basis=basis, diag_tol=diag_tol, ortho_center=cut, device=device, as_tenpy=False)
    res, err = iMPS.MPS_to_iMPS(mps_s, mps_l, sites_per_cell, cut, unitary_tol=unitary_tol, schmidt_tol=schmidt_tol,
                                offset=0, unit_cell_width=sites_per_cell, device=device, right="project")
    res.unit_cell_width = unit_cell_width
    from .slater import _maybe_tenpy

    return _maybe_tenpy(res, as_tenpy), err


def H_to_iMPS(H_short: np.ndarray, H_long: np.ndarray, trunc_par: dict | StoppingCondition, sites_per_cell: int,
              cut: int, *, basis: str, diag_tol: float = _DIAG_TOL, unitary_tol: float = 1e-6,
              schmidt_tol: float = 1e-6, unit_cell_width: int | None = None, device: str = "cuda:0",
              as_tenpy: bool | None = None):
    """pfaffian.py:2151-2242."""
    C_short = correlation_matrix(H_short, basis=f"{basis}->{basis}")
    C_long = correlation_matrix(H_long, basis=f"{basis}->{basis}")
    return C_to_iMPS(C_short, C_long, trunc_par, sites_per_cell, cut, basis=basis, diag_tol=diag_tol,
                     unitary_tol=unitary_tol, schmidt_tol=schmidt_tol, unit_cell_width=unit_cell_width, device=device,
                     as_tenpy=as_tenpy)
