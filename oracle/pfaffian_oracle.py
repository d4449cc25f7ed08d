"""CPU oracle for the Pfaffian (BCS / Nambu mean-field) -> MPS sweep.  TEST INFRASTRUCTURE ONLY.

NumPy restatement of ``/root/reference/src/temfpy/pfaffian.py`` (citations are file:line in that
tree), pinned by ``tests/test_oracle_pfaffian.py`` against fixtures the reference's own NumPy core
produced (``tests/golden/make_golden_pfaffian.py``).

Parity unpinned (third-party code that is not installed anywhere we can run):
  * ``pfapack.ctypes.pfaffian`` (pfaffian.py:49,1425; version unpinned in pyproject.toml:37): the
    fixtures were generated with the Parlett-Reid routine below in its place; it is validated by
    Pf(A)^2 = det(A), Pf(B A B^T) = det(B) Pf(A) and closed forms (tests).
  * TeNPy's LegCharge / LegPipe bookkeeping (pfaffian.py:1485-1489, 1655-1657, 1758-1778).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.stats import ortho_group

from .slater_oracle import Trunc, as_trunc, lowest_sums


# --------------------------------------------------------------------------------------
# Pfaffian of a skew-symmetric matrix (stand-in for pfapack's skpfa, Parlett-Reid)
# --------------------------------------------------------------------------------------
def pfaffian(A) -> complex:
    A = np.array(A, dtype=complex if np.iscomplexobj(A) else float)
    n = len(A)
    if n % 2:
        return 0.0
    pf = 1.0
    for k in range(0, n - 1, 2):
        kp = k + 1 + int(np.argmax(np.abs(A[k, k + 1:])))
        if kp != k + 1:
            A[[k + 1, kp]] = A[[kp, k + 1]]
            A[:, [k + 1, kp]] = A[:, [kp, k + 1]]
            pf = -pf
        piv = A[k, k + 1]
        if piv == 0:
            return 0.0
        pf = pf * piv
        if k + 2 < n:
            tau = A[k, k + 2:] / piv
            col = A[k + 2:, k + 1]
            A[k + 2:, k + 2:] += np.outer(tau, col) - np.outer(col, tau)
    return pf


# --------------------------------------------------------------------------------------
# basis changes (pfaffian.py:75-184)
# --------------------------------------------------------------------------------------
_M2C = np.array([[1, -1j], [1, 1j]]) / 2**0.5   # pfaffian.py:122
_C2M = np.array([[1, 1], [1j, -1j]]) / 2**0.5   # pfaffian.py:93


def vector_M2C(v):
    n = v.shape[0] // 2
    w = v.reshape(n, 2, *v.shape[1:])
    return np.einsum("xa...,ca->xc...", w, _M2C).reshape(2 * n, *v.shape[1:])


def matrix_M2C(H):
    n, m = H.shape[0] // 2, H.shape[1] // 2
    return np.einsum("xayb,ca,db->xcyd", H.reshape(n, 2, m, 2), _M2C, _M2C.conj()).reshape(2 * n, 2 * m)


def matrix_C2M(H):
    n, m = H.shape[0] // 2, H.shape[1] // 2
    return np.einsum("xayb,ca,db->xcyd", H.reshape(n, 2, m, 2), _C2M, _C2M.conj()).reshape(2 * n, 2 * m)


def correlation_matrix(H_majorana, atol=1e-10):
    """pfaffian.py:302-393 for basis "M->M"."""
    H = (H_majorana + H_majorana.conj().T) / 2
    H = 1j * H.imag  # pfaffian.py:269-273 (offset 0: the real part must vanish)
    n = len(H) // 2
    e, v = np.linalg.eigh(H)
    if np.any(np.abs(e) < atol):
        raise RuntimeError("Some energy eigenvalues are zero.")  # pfaffian.py:372-377
    v = v[:, :n]
    C = v @ v.conj().T
    C = (C + C.conj().T) / 2
    C = 0.5 * np.eye(2 * n) + 1j * C.imag  # Majorana basis: real part = 1/2 (pfaffian.py:271-273)
    return C


def vacuum_parity(V, tol=1e-12) -> int:
    """pfaffian.py:396-456: parity of a Bogoliubov vacuum from the singular values of V."""
    if len(V) == 0:
        return 0
    if len(V) == 1:
        v = abs(V.item())
        if np.isclose(v, 0.0, rtol=0, atol=tol):
            return 0
        if np.isclose(v, 1.0, rtol=0, atol=tol):
            return 1
        raise RuntimeError("Invalid 1x1 V")
    s = np.linalg.svd(V, compute_uv=False)
    if len(V) > 2:
        return (int(np.argmax(-np.diff(s))) + 1) % 2
    if np.allclose(s, [1.0, 0.0], rtol=0, atol=tol):
        return 1
    if np.isclose(s[0], s[1], rtol=0, atol=tol):
        return 0
    raise ValueError("Invalid 2x2 V")


# --------------------------------------------------------------------------------------
# one entanglement cut
# --------------------------------------------------------------------------------------
@dataclass
class PCut:
    x: int
    nL: int
    nR: int
    e: np.ndarray            # entangled eigenvalues, ascending, in [cutoff, 1/2]
    vL: np.ndarray | None    # (2 nL, 2 nL) Bogoliubov matrix, complex-fermion rows, [a | a^dag] columns
    vR: np.ndarray | None
    pL: int | None
    pR: int | None
    kh: int = 0
    sets: np.ndarray = None  # (chi, k) bool: excited gamma^dag modes (left order)
    lam_raw: np.ndarray = None
    lam: np.ndarray = None
    idx_n: dict = field(default_factory=dict)       # n_exc -> (start, stop)
    idx_parity: dict = field(default_factory=dict)  # parity -> (start, stop)

    @property
    def k(self):
        return self.e.size

    def parity(self, which="T"):
        if which == "L":
            return self.pL
        if which == "R":
            return self.pR
        return None if (self.pL is None or self.pR is None) else (self.pL + self.pR) % 2

    def side_sets(self, side):
        return self.sets if side == "L" else self.sets[:, ::-1]  # pfaffian.py:955-957


def _block_svd(CLR, vL, vR, e, tol):
    """utils.py:19-96 (same routine as in the Slater path)."""
    k = e.size
    if k == 0:
        return
    brk = np.nonzero(np.abs(np.diff(e)) > tol)[0] + 1
    starts = np.concatenate(([0], brk))
    mult = np.diff(np.concatenate((starts, [k])))
    for m in np.unique(mult):
        ix = starts[mult == m, None] + np.arange(m)
        blk = np.einsum("kdi,km,mdj->dij", vL[:, ix].conj(), CLR, vR[:, ix])
        U, _, Vh = np.linalg.svd(blk)
        vL[:, ix] = np.einsum("idk,dkj->idj", vL[:, ix], U)
        vR[:, ix] = np.einsum("idk,djk->idj", vR[:, ix], Vh.conj())


def _diag_nambu(c, cutoff, deg_tol, diag_tol):
    """pfaffian.py:764-823."""
    n = len(c) // 2
    if n == 0:
        return np.zeros(0), np.zeros((0, 0), c.dtype), 0, 0
    e, v = np.linalg.eigh(c)
    e = np.clip(e, 0.0, 1.0)
    x0, x1 = np.searchsorted(e, [0.5 - deg_tol, 0.5 + deg_tol])
    kh = x1 - n
    assert x0 == n - kh
    if kh != 0 and np.iscomplexobj(v):  # make the 1/2 modes real (pfaffian.py:807-816)
        w = np.column_stack((v[:, x0:x1].real, v[:, x0:x1].imag))
        w, _, _ = np.linalg.svd(w)
        v[:, x0:x1] = w[:, : 2 * kh]
    x0, x1 = np.searchsorted(e, [cutoff, 1 - cutoff])
    ke = x1 - n
    assert x0 == n - ke
    return e, v, ke, kh


def _nambu(v, kh, side):
    """pfaffian.py:880-897: restore conjugate pairs, go to the complex-fermion basis, vacuum parity."""
    x = len(v) // 2
    if side == "L":
        v[:, x - kh: x] = (v[:, x - kh: x] + 1j * v[:, x: x + kh]) / 2**0.5
        v[:, x:] = v[:, :x].conj()
    else:
        v[:, x: x + kh] = (-1j * v[:, x - kh: x] + v[:, x: x + kh]) / 2**0.5
        v[:, x: x + kh] = v[:, x: x + kh][:, ::-1]
        v[:, :x] = v[:, x:].conj()
    v = vector_M2C(v)
    return v, vacuum_parity(v[1::2, :x])


def cut_modes(C, x, trunc: Trunc, which="LR", total_parity=None, diag_tol=1e-8) -> PCut:
    """pfaffian.py:685-920 (C in the Majorana basis)."""
    cutoff, deg_tol = trunc.svd_min**2, trunc.degeneracy_tol
    C = (C + C.conj().T) / 2
    C = 0.5 * np.eye(len(C)) + 1j * C.imag  # assert_nambu_correlation(C, "M"), pfaffian.py:754
    L = len(C) // 2
    y = L - x
    eL = vL = eR = vR = None
    if "L" in which:
        eL, vL, keL, khL = _diag_nambu(C[: 2 * x, : 2 * x], cutoff, deg_tol, diag_tol)
    if "R" in which:
        eR, vR, keR, khR = _diag_nambu(C[2 * x:, 2 * x:], cutoff, deg_tol, diag_tol)
    if eL is None:
        k, kh, e = keR, khR, eR[y - keR: y]
    elif eR is None:
        k, kh, e = keL, khL, eL[x - keL: x]
    else:
        assert keL == keR and khL == khR
        k, kh, e = keL, khL, eL[x - keL: x]
        CLR = C[: 2 * x, 2 * x:]
        _block_svd(CLR, vL[:, x - k: x - kh], vR[:, y + kh: y + k][:, ::-1], eL[x - k: x - kh], deg_tol)
        if kh:  # pfaffian.py:857-865
            ixL, ixR = slice(x - kh, x + kh), slice(y - kh, y + kh)
            U, _, Vh = np.linalg.svd(vL[:, ixL].real.T @ CLR.imag @ vR[:, ixR].real)
            vL[:, ixL] = vL[:, ixL] @ U
            vR[:, ixR] = vR[:, ixR] @ Vh.T
    if kh > 0:  # fixed-seed shuffle of the 1/2 modes (pfaffian.py:867-874)
        O = ortho_group.rvs(2 * kh, random_state=1234)
        if vL is not None:
            vL[:, x - kh: x + kh] = vL[:, x - kh: x + kh] @ O
        if vR is not None:
            vR[:, y - kh: y + kh] = vR[:, y - kh: y + kh] @ O
    pL = pR = None
    if "L" in which:
        vL, pL = _nambu(vL, kh, "L")
        if "R" not in which and total_parity is not None:
            pR = (total_parity + pL) % 2
    if "R" in which:
        vR, pR = _nambu(vR, kh, "R")
        if "L" not in which and total_parity is not None:
            pL = (total_parity + pR) % 2
    if "L" in which and "R" in which and pL == 1:
        vR = -vR  # pfaffian.py:915-916
    return PCut(x=x, nL=x, nR=y, e=e, vL=vL, vR=vR, pL=pL, pR=pR, kh=kh)


def _bunched(x):
    idx = np.nonzero(x[1:] != x[:-1])[0]
    idx = np.concatenate(([0], idx + 1, [len(x)]))
    return {int(x[idx[i]]): (int(idx[i]), int(idx[i + 1])) for i in range(len(idx) - 1)}


def parity_n_argsort(exc):
    """pfaffian.py:986-997: sort by parity, then number, then original order."""
    exc = np.asarray(exc).ravel()
    idx = np.lexsort((np.arange(len(exc)), exc, exc % 2))
    s = exc[idx]
    return idx, _bunched(s), _bunched(s % 2)


def cut_vectors(C, x, trunc: Trunc, which="LR", total_parity=None) -> PCut:
    """pfaffian.py:1162-1248."""
    cut = cut_modes(C, x, trunc, which, total_parity)
    a = np.log((1 - cut.e) / cut.e) / 2
    _, sets, _ = lowest_sums(a, trunc)
    if len(sets) == 0:
        raise ValueError("No Schmidt vectors left after filtering by `trunc_par.sectors`!")
    idx, cut.idx_n, cut.idx_parity = parity_n_argsort(sets.sum(axis=1))
    cut.sets = sets[idx]
    cut.lam_raw = np.where(cut.sets, cut.e, 1 - cut.e).prod(axis=1) ** 0.5
    cut.lam = cut.lam_raw / np.linalg.norm(cut.lam_raw)
    return cut


# --------------------------------------------------------------------------------------
# one site tensor
# --------------------------------------------------------------------------------------
@dataclass
class PSite:
    mode: str
    norm: float
    N: np.ndarray            # skew-symmetric Pfaffian matrix [[BB, BA], [-BA^T, AA]]
    sets_bra: np.ndarray     # rows sorted by (parity, number); columns = [ket-mode zeros | bra modes]
    sets_ket: np.ndarray
    leg_idx_bra: np.ndarray  # position of every sorted bra row in the merged (p, bra) leg
    qtotal: int
    idx_n_bra: dict
    idx_n_ket: dict
    blocks: dict             # (n_bra, n_ket) -> (r0, r1, c0, c1, block)


def pfaffian_matrix(V1, V2, sets1, sets2, mode, min_SV=1e-6):
    """pfaffian.py:1258-1410."""
    L = len(V1) // 2
    Vr = V1.conj().T @ V2
    s = np.linalg.svd(Vr[:L, :L], compute_uv=False)
    norm = s.prod() ** 0.5  # Onishi formula (pfaffian.py:1359)

    def prune(sets, reverse):
        idx = np.nonzero(np.any(sets, axis=0))[0]
        if reverse:
            idx = idx[::-1]
        return sets[:, idx], idx

    act1, act2 = sets1.shape[1], sets2.shape[1]
    sets1, idx1 = prune(sets1, False)
    sets2, idx2 = prune(sets2, True)
    if mode == "left":  # active modes at the end (pfaffian.py:1377-1379)
        idx1 = idx1 + L - act1
        idx2 = idx2 + L - act2
    Uxinv = np.linalg.inv(Vr[L:, L:])
    AA = Vr[idx1, L:] @ Uxinv[:, idx1]
    BA = Uxinv[np.ix_(idx2, idx1)]
    BB = Uxinv[idx2] @ Vr[L:, idx2]
    AA = (AA - AA.T) / 2
    BB = (BB - BB.T) / 2
    N = np.block([[BB, BA], [-BA.T, AA]])
    new1 = np.concatenate((np.zeros((len(sets1), sets2.shape[1]), bool), sets1), axis=1)
    new2 = np.concatenate((sets2, np.zeros((len(sets2), sets1.shape[1]), bool)), axis=1)
    return norm, N, new1, new2


def batched_sub_pfaffians(N, s1, s2):
    """pfaffian.py:1429-1479: Pf of N restricted to [ket positions, bra positions] for all pairs."""
    n1, n2 = int(s1[0].sum()), int(s2[0].sum())
    assert np.all(s1.sum(axis=1) == n1) and np.all(s2.sum(axis=1) == n2) and (n1 + n2) % 2 == 0
    i1 = np.nonzero(s1)[1].reshape(len(s1), n1)
    i2 = np.nonzero(s2)[1].reshape(len(s2), n2)
    out = np.zeros((len(s1), len(s2)), N.dtype)
    for a in range(len(s1)):
        for b in range(len(s2)):
            ix = np.concatenate((i2[b], i1[a]))
            out[a, b] = pfaffian(N[np.ix_(ix, ix)])
    return out


def site_tensor(bra: PCut, ket: PCut, mode: str) -> PSite:
    """pfaffian.py:1578-1748 and the block loop of pfaffian.py:1766-1776."""
    side = "L" if mode == "left" else "R"
    v_bra = (bra.vL if side == "L" else bra.vR).copy()
    v_ket = ket.vL if side == "L" else ket.vR
    sets_bra = bra.side_sets(side).copy()
    if bra.pL is None or ket.pL is None:
        qtotal = 0
    elif mode == "right":
        qtotal = (bra.parity() + ket.parity()) % 2
    else:
        qtotal = 0
    assert len(v_bra) + 2 == len(v_ket), "bra must be one site shorter than ket"
    ns, n = len(sets_bra), len(v_bra) // 2
    zc, zr = np.zeros((2 * n, 1)), np.zeros((1, n))
    z, o = np.zeros((ns, 1), bool), np.ones((ns, 1), bool)
    if mode == "left":  # pfaffian.py:1662-1679
        u_p = -1 if bra.parity("L") % 2 == 1 else 1
        v_bra = np.block([[v_bra[:, :n], zc, v_bra[:, n:], zc], [zr, u_p, zr, 0.0], [zr, 0.0, zr, u_p]])
        sets_bra = np.block([[sets_bra, z], [sets_bra, o]])
    else:  # pfaffian.py:1680-1694
        v_bra = np.block([[1, zr, 0, zr], [0, zr, 1, zr], [zc, v_bra[:, :n], zc, v_bra[:, n:]]])
        sets_bra = np.block([[z, sets_bra], [o, sets_bra]])
    if bra.parity(side) % 2 != ket.parity(side) % 2:  # reference-parity fix (pfaffian.py:1707-1719)
        n = len(v_bra) // 2
        if mode == "left":
            v_bra[:, [n - 1, -1]] = v_bra[:, [-1, n - 1]]
            sets_bra[:, -1] = ~sets_bra[:, -1]
        else:
            v_bra = -v_bra
            v_bra[:, [0, n]] = -v_bra[:, [n, 0]]
            sets_bra[:, 0] = ~sets_bra[:, 0]
    norm, N, sb, sk = pfaffian_matrix(v_bra, v_ket, sets_bra, ket.side_sets(side), mode)
    leg_idx, idx_n_bra, _ = parity_n_argsort(sb.sum(axis=1))
    sb = sb[leg_idx]
    blocks = {}
    for nb, (r0, r1) in idx_n_bra.items():
        for nk, (c0, c1) in ket.idx_n.items():
            if (nb + nk) % 2 == 1:
                continue
            blocks[(nb, nk)] = (r0, r1, c0, c1, norm * batched_sub_pfaffians(N, sb[r0:r1], sk[c0:c1]))
    return PSite(mode, norm, N, sb, sk, leg_idx, qtotal, idx_n_bra, dict(ket.idx_n), blocks)


def c_to_mps(C, trunc, ortho_center=None):
    """pfaffian.py:1785-1921 (basis "M") without the TeNPy container: returns (cuts, sites)."""
    trunc = as_trunc(trunc)
    L = len(C) // 2
    oc = ortho_center or L // 2
    cuts, sites = [None] * (L + 1), [None] * L
    cuts[oc] = cut_vectors(C, oc, trunc, "LR")
    parity = cuts[oc].parity()
    for i in range(oc, L):
        cuts[i + 1] = cut_vectors(C, i + 1, trunc, "R", parity)
        sites[i] = site_tensor(cuts[i + 1], cuts[i], "right")
    for i in reversed(range(oc)):
        cuts[i] = cut_vectors(C, i, trunc, "L", parity)
        sites[i] = site_tensor(cuts[i], cuts[i + 1], "left")
    return cuts, sites
