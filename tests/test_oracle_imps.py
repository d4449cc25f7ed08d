"""CPU tests of the iMPS oracle (oracle/imps_oracle.py).  TeNPy is not installed anywhere we can run, so the
restatement is pinned by the acceptance check of the reference's own example (src/examples/iMPS.py:27-38):
short chain + n inserted unit cells of the iMPS must reproduce the directly converted longer chain."""
import numpy as np
import pytest

from oracle import imps_oracle as io
from oracle import slater_oracle as orc


def ssh(L, t1=-1.5, t2=-1.0):
    """src/examples/iMPS.py:6-10 with the strong bond first (no edge modes, so that L = 32 suffices)."""
    M = t1 * np.ones(L - 1)
    M[1::2] = t2
    M = np.diag(M, 1)
    return M + M.T


def finite(L, chi, oc=None):
    C, _ = orc.correlation_matrix(ssh(L))
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi}, ortho_center=oc)
    oc = oc or L // 2
    return orc.dense_tensors(cuts, sites), [c.lam for c in cuts], ["A"] * oc + ["B"] * (L - oc)


@pytest.mark.parametrize("L,cut,oc_long", [(32, 16, None), (32, 16, 16), (28, 12, None)])
def test_reconstruction_overlap(L, cut, oc_long):
    chi = 48
    Ts, ls, fs = finite(L, chi, cut if cut != L // 2 else None)
    Tl, ll, fl = finite(L + 2, chi, oc_long)
    B, S, err = io.mps_to_imps(Ts, ls, fs, Tl, ll, fl, 2, cut)
    assert max(err) < 2e-4          # the chi-truncated chains differ at the svd_min = 1e-6 level (/ lam when forms are converted)
    for t, sl in zip(B, S):          # right-canonical up to the truncation, weighted with the Schmidt values
        X = sum(t[p] @ t[p].conj().T for p in range(2)) - np.eye(t.shape[1])
        assert np.abs(sl[:, None] * X * sl[None, :]).max() < 1e-6
    n_cell = 3
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, B, S, cut, n_cell)
    Tv, lv, fv = finite(L + 2 * n_cell, chi, cut if cut != L // 2 else None)
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr = io.overlap(Tr, lr, fr, Tr, lr, fr).real
    nv = io.overlap(Tv, lv, fv, Tv, lv, fv).real
    assert abs(abs(ov) / np.sqrt(nr * nv) - 1) < 1e-8


def test_basis_rotation_is_the_polar_factor():
    rng = np.random.default_rng(0)
    n = 7
    U = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))[0]
    S = np.sort(rng.uniform(0.1, 1, n))[::-1]
    S /= np.linalg.norm(S)
    rot, ue, se = io.basis_rotation(U, S, S, "left")       # Procrustes of U S^2: the unitary overlap itself
    np.testing.assert_allclose(rot, U, atol=1e-12)
    assert ue < 1e-7
    Dg = np.diag(np.exp(1j * rng.uniform(0, 6, n)))          # commutes with the Schmidt values: no mixing
    for mode in ("left", "right"):
        rot, ue, se = io.basis_rotation(Dg, S, S, mode)
        np.testing.assert_allclose(rot, Dg, atol=1e-12)
        assert ue < 1e-7 and se < 1e-12
    rot, ue, se = io.basis_rotation(0.9 * U, S, S, "left")
    np.testing.assert_allclose(rot, U, atol=1e-12)
    assert abs(ue - np.sqrt(1 - 0.81)) < 1e-12


@pytest.mark.parametrize("L,cut,cell", [(32, 16, 2), (28, 12, 2), (30, 14, 4)])
def test_c_to_imps_by_determinant_formulas(L, cut, cell):
    """slater.py:1356-1565 restated (oracle ``c_to_imps``): every tensor is a determinant overlap between Schmidt vectors of
    two cuts - the gauge matrix without a physical leg (slater.py:1023-1024), the last tensor with the right vectors of the
    SHORT chain as bra (:1513-1514) - and no chain is converted in full.  Pinned like the rest of this oracle: the short chain
    with n cells inserted reproduces the directly converted longer chain; and the overlaps agree with those of the
    transfer-matrix route (``mps_to_imps``), weighted with the Schmidt values of the two bases."""
    chi = 48
    Cs, _ = orc.correlation_matrix(ssh(L))
    Cl, _ = orc.correlation_matrix(ssh(L + cell))
    B, S, (lu, ls_), G = orc.c_to_imps(Cs, Cl, {"chi_max": chi}, cell, cut)
    assert lu < 2e-4 and ls_ < 2e-4
    assert len(B) == cell and len(S) == cell + 1 and S[0] is S[-1]
    for t, sl in zip(B, S):
        X = sum(t[p] @ t[p].conj().T for p in range(2)) - np.eye(t.shape[1])
        assert np.abs(sl[:, None] * X * sl[None, :]).max() < 1e-6
    Ts, ls, fs = finite(L, chi, cut if cut != L // 2 else None)
    n_cell = 3
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, B, S, cut, n_cell)
    Tv, lv, fv = finite(L + cell * n_cell, chi, cut if cut != L // 2 else None)
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr = io.overlap(Tr, lr, fr, Tr, lr, fr).real
    nv = io.overlap(Tv, lv, fv, Tv, lv, fv).real
    assert abs(abs(ov) / np.sqrt(nr * nv) - 1) < 1e-8
    # the same gauge overlaps from the transfer matrices of the two fully converted chains
    Tl, ll, fl = finite(L + cell, chi, cut)
    bra = [io.get_B(Ts, ls, fs, i, "A") for i in range(cut)]
    ket = [io.get_B(Tl, ll, fl, i, "A") for i in range(cut)]
    w = S[0][:, None] * ll[cut][None, :]      # (entries between weakly weighted vectors see the truncation of the chains)
    assert (np.abs(io.overlap_schmidt(bra, ket, "left") - G) * w).max() < 1e-9
