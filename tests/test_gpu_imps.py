"""GPU parity tests of the finite -> infinite MPS conversion (temfpy_amd/iMPS.py, HIP path through the C ABI)
against the oracle (oracle/imps_oracle.py) fed with the SAME finite MPS, and the acceptance check of the
reference's example (src/examples/iMPS.py:27-38).  Tolerances (fp64): overlaps 1e-10, rotations 1e-9 after
weighting with the Schmidt values (directions of weight < 1e-6 are fixed by rounding only), error metrics
1e-9, unit-cell tensors 1e-9 weighted, reconstruction overlap 1 - 1e-8."""
import warnings

import numpy as np
import pytest

from oracle import imps_oracle as io

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def ssh(L, t1=-1.5, t2=-1.0, imag=0.0):
    M = t1 * np.ones(L - 1, complex if imag else float)
    M[1::2] = t2
    if imag:
        M = M * np.exp(1j * imag)        # uniform Peierls phase: complex tensors, still translation invariant
    M = np.diag(M, 1)
    return M + M.conj().T


def finite(L, chi, oc=None, imag=0.0):
    from temfpy_amd import slater

    C, _ = slater.correlation_matrix(ssh(L, imag=imag))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return slater.C_to_MPS(C, {"chi_max": chi}, ortho_center=oc, as_tenpy=False)


def dense(m):
    return m.dense_tensors(), [np.asarray(x) for x in m.lam], list(m.form)


@pytest.mark.parametrize("L,cut,oc_long,imag", [(32, 16, None, 0.0), (32, 16, 16, 0.0), (28, 12, None, 0.0),
                                                (32, 16, None, 0.3)])
def test_mps_to_imps_against_oracle(L, cut, oc_long, imag):
    from temfpy_amd import iMPS

    chi = 48
    ms = finite(L, chi, cut if cut != L // 2 else None, imag)
    ml = finite(L + 2, chi, oc_long, imag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, err = iMPS.MPS_to_iMPS(ms, ml, 2, cut, offset=0)
    Ts, ls, fs = dense(ms)
    Tl, ll, fl = dense(ml)
    B, S, eo = io.mps_to_imps(Ts, ls, fs, Tl, ll, fl, 2, cut)
    np.testing.assert_allclose(list(err), eo, rtol=0, atol=1e-9)
    assert isinstance(err, iMPS.iMPSError) and abs(err.total_error - np.linalg.norm(eo)) < 1e-9
    assert res.L == 2 and res.form == ["B", "B"] and res.bc == "infinite"
    for a, b in zip(res.lam, S):
        np.testing.assert_allclose(a, b, atol=0)
    for t, r, sl, sr in zip(res.dense_tensors(), B, S[:-1], S[1:]):
        w = sl[None, :, None] * np.abs(t - r) * 1.0
        assert w.max() < 1e-9, w.max()
    # acceptance check of the reference's example: short chain + n unit cells = the longer chain
    n_cell = 3
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, cut, n_cell)
    Tv, lv, fv = dense(finite(L + 2 * n_cell, chi, cut if cut != L // 2 else None, imag))
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr, nv = io.overlap(Tr, lr, fr, Tr, lr, fr).real, io.overlap(Tv, lv, fv, Tv, lv, fv).real
    assert abs(abs(ov) / np.sqrt(nr * nv) - 1) < 1e-8


def test_overlap_and_rotation_entry_points():
    from temfpy_amd import iMPS

    ms, ml = finite(24, 32, 12), finite(26, 32, 12)
    Ts, ls, fs = dense(ms)
    Tl, ll, fl = dense(ml)
    C0 = iMPS.overlap_schmidt(ms, ml, "left", segment_bra=(0, 12), segment_ket=(0, 12))
    ref = io.overlap_schmidt([io.get_B(Ts, ls, fs, i, "A") for i in range(12)],
                             [io.get_B(Tl, ll, fl, i, "A") for i in range(12)], "left")
    np.testing.assert_allclose(C0.dense(), ref, atol=1e-10)
    D0 = iMPS.overlap_schmidt(ms, ml, "right", segment_bra=(12, 24), segment_ket=(14, 26))
    ref = io.overlap_schmidt([io.get_B(Ts, ls, fs, i, "B") for i in range(12, 24)],
                             [io.get_B(Tl, ll, fl, i, "B") for i in range(14, 26)], "right")
    np.testing.assert_allclose(D0.dense(), ref, atol=1e-10)
    for mode, ov, Sb, Sk in (("left", C0, ls[12], ll[12]), ("right", D0, ls[12], ll[14])):
        for form in ("A", "B"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                rot, ue, se = iMPS.basis_rotation(ov, Sb, Sk, mode, form=form)
            r0, u0, s0 = io.basis_rotation(ov.dense(), Sb, Sk, mode, form)
            assert abs(ue - u0) < 1e-10 and abs(se - s0) < 1e-9
            w = (Sb[:, None] if mode == "left" else Sk[:, None]) * np.abs(rot.dense() - r0)
            assert w.max() < 1e-9
    with pytest.raises(ValueError, match="mode"):
        iMPS.overlap_schmidt(ms, ml, "up")
    with pytest.raises(ValueError, match="differ by one unit cell"):
        iMPS.MPS_to_iMPS(ms, ml, 4, 12)
    with pytest.warns(UserWarning, match="deviates from unitarity"):
        iMPS.basis_rotation(C0, ls[12], ll[12], "left", unitary_tol=1e-12)


@pytest.mark.parametrize("spinful", [None, "simple", "PH"])
def test_H_to_iMPS(spinful):
    """slater.H_to_iMPS (slater.py:1630-1734): unit cell of the SSH chain; offsets as the reference defines
    them; inserting cells into the short chain reproduces the longer chain."""
    from temfpy_amd import slater

    L, cut, chi, n_cell = 24, 12, 40 if spinful is None else 600, 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, err = slater.H_to_iMPS(ssh(L), ssh(L + 2), {"chi_max": chi}, 2, cut, spinful=spinful)
        mult = 1 if spinful is None else 2
        assert res.L == 2 * mult and res.unit_cell_width == 2 and err.total_error < 1e-3
        assert err.right_unitary == 0.0 and err.right_schmidt == 0.0     # slater.py:1563
        expect = {None: 6, "simple": 12, "PH": 12}[spinful]          # particles left of the cut at half filling
        q0 = res.charges[0] + expect
        Cs, _ = slater.correlation_matrix(ssh(L))
        ms = slater.C_to_MPS(Cs, {"chi_max": chi}, ortho_center=mult * cut, spinful=spinful, as_tenpy=False)
        assert np.array_equal(q0, np.asarray(ms.bonds[mult * cut].q_left))
        Cv, _ = slater.correlation_matrix(ssh(L + 2 * n_cell))
        mv = slater.C_to_MPS(Cv, {"chi_max": chi}, ortho_center=mult * cut, spinful=spinful, as_tenpy=False)
    Ts, ls, fs = dense(ms)
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, mult * cut, n_cell)
    Tv, lv, fv = dense(mv)
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr, nv = io.overlap(Tr, lr, fr, Tr, lr, fr).real, io.overlap(Tv, lv, fv, Tv, lv, fv).real
    assert abs(abs(ov) / np.sqrt(nr * nv) - 1) < 1e-6
    with pytest.raises(ValueError, match="does not divide"):
        slater.H_to_iMPS(ssh(L), ssh(L + 2), {"chi_max": chi}, 2, cut, unit_cell_width=3)
    with pytest.raises(ValueError, match="spinful"):
        slater.H_to_iMPS(ssh(L), ssh(L + 2), {"chi_max": chi}, 2, cut, spinful="both")


@pytest.mark.parametrize("L,cut,cell,imag", [(32, 16, 2, 0.0), (28, 12, 2, 0.0), (30, 14, 4, 0.0), (32, 16, 2, 0.3), (26, 13, 2, 0.0)])
def test_C_to_iMPS_is_the_reference_s_determinant_construction(L, cut, cell, imag, monkeypatch):
    """slater.C_to_iMPS against the oracle's restatement of slater.py:1499-1563 (``slater_oracle.c_to_imps``): the overlaps
    of the left Schmidt vectors of the two chains (no physical leg) to 1e-10 in modulus, Schmidt-weighted (the sign of a Schmidt vector is a
    product of orbital phases; vectors of equal weight may change places), Schmidt values 1e-9, error metrics 1e-9, and the unit cells describe the same state: the
    short chain with three cells of either inserted, overlap 1 to 1e-9.  Neither chain is converted in full (the engine's
    full-conversion entry point is not called)."""
    from oracle import slater_oracle as orc
    from temfpy_amd import slater
    from temfpy_amd.engine import Engine

    chi = 48
    Cs, _ = slater.correlation_matrix(ssh(L, imag=imag))
    Cl, _ = slater.correlation_matrix(ssh(L + cell, imag=imag))
    To, So, (lu, ls_), G = orc.c_to_imps(Cs, Cl, {"chi_max": chi}, cell, cut)
    calls = []
    orig = Engine.run
    monkeypatch.setattr(Engine, "run", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, err = slater.C_to_iMPS(Cs, Cl, {"chi_max": chi}, cell, cut, as_tenpy=False)
    assert not calls, "C_to_iMPS converted a whole chain"
    monkeypatch.undo()
    assert res.L == cell and len(res.lam) == cell + 1
    for a, b in zip(res.lam, So):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)       # (values at the 1e-6 cutoff come from eigenvalues ~1e-12)
    Gd = res.gauge_overlaps.dense()
    assert Gd.shape == G.shape
    # Schmidt vectors with a partner of the same weight (particle-hole symmetric chain: a(e) = -a(1 - e)) come in either
    # order at double precision; entries between the others are compared one by one, the blocks by their singular values
    def single(lam):
        lam = np.asarray(lam)
        return np.array([np.min(np.abs(np.delete(lam, i) / x - 1), initial=1.0) > 1e-6 for i, x in enumerate(lam)])
    from temfpy_amd.iMPS import _sector_table
    rows_ok = single(res.lam[0])
    # (weighted with the Schmidt values of the two bases, as the construction uses them, iMPS.py:135: a vector of weight
    # 1e-5 comes from an eigenvalue ~1e-10, which C fixes to ~1e-6 only)
    w = np.asarray(So[0])[:, None] * np.asarray(So[0])[None, :]
    if G.shape[0] == G.shape[1]:     # same truncated dimension on both chains: the partner structure is the same
        sub = np.ix_(rows_ok, rows_ok)
        np.testing.assert_allclose((w * np.abs(Gd))[sub], (w * np.abs(G))[sub], rtol=0, atol=1e-10)
        assert rows_ok.sum() >= 4 or cut % 2 == 1     # (a cut through a strong bond: every Schmidt value is twofold degenerate)
    rt = _sector_table(res.gauge_overlaps.rows)
    ct = _sector_table(res.gauge_overlaps.cols)
    for (qr, qc) in res.gauge_overlaps.blocks:
        sl_ = (slice(rt[qr][0], rt[qr][0] + rt[qr][1]), slice(ct[qc][0], ct[qc][0] + ct[qc][1]))
        wa = np.asarray(res.lam[0])[sl_[0], None] * Gd[sl_] * np.asarray(So[0])[None, sl_[1]] if G.shape[0] == G.shape[1] else Gd[sl_]
        wb = np.asarray(So[0])[sl_[0], None] * G[sl_] * np.asarray(So[0])[None, sl_[1]] if G.shape[0] == G.shape[1] else G[sl_]
        np.testing.assert_allclose(np.linalg.svd(wa, compute_uv=False), np.linalg.svd(wb, compute_uv=False), rtol=0,
                                   atol=1e-10 if G.shape[0] == G.shape[1] else 1e-6)
    # (the unitarity defect is the root of a difference of two numbers ~1: compared as the square)
    assert abs(err.left_unitary**2 - lu**2) < 1e-13
    # (Schmidt value mixing: the polar factor is fixed by rounding in directions of weight ~1e-6, which enter it with that
    # weight - agreement to 1e-9 when the overlaps are unitary to that level, else to a factor of two)
    assert abs(err.left_schmidt - ls_) < 1e-9 or (ls_ > 1e-7 and 0.5 < err.left_schmidt / ls_ < 2.0)
    assert err.right_unitary == 0.0 and err.right_schmidt == 0.0
    # same infinite state, whatever the gauge of the bond bases: the mixed transfer matrix of the two unit cells has a
    # dominant eigenvalue of modulus 1
    Ta = res.dense_tensors()
    E = None
    for a_, b_ in zip(Ta, To):
        step = np.einsum("pab,pcd->acbd", a_.conj(), b_)          # (a, c) -> (b, d)
        step = step.reshape(a_.shape[1] * b_.shape[1], a_.shape[2] * b_.shape[2])
        E = step if E is None else E @ step
    eta = np.abs(np.linalg.eigvals(E)).max()
    assert abs(eta - 1) < 1e-8, eta
    # acceptance check of src/examples/iMPS.py:27-38 with a SEPARATELY converted short chain: the cell comes in the same
    # gauge as ``C_to_MPS(C_short, ortho_center=cut)`` (canonical phases of the entangled orbitals)
    ms = finite(L, chi, cut, imag=imag)
    Ts, ls, fs = dense(ms)
    n_cell = 3
    Ta, la, fa = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, cut, n_cell)
    na = io.overlap(Ta, la, fa, Ta, la, fa).real
    mv = finite(L + cell * n_cell, chi, cut, imag=imag)
    Tv, lv, fv = dense(mv)
    ov = io.overlap(Tv, lv, fv, Ta, la, fa)
    assert abs(abs(ov) / np.sqrt(na * io.overlap(Tv, lv, fv, Tv, lv, fv).real) - 1) < 1e-8


def kitaev(L, t1=1.5j, t2=1j):
    """Majorana Hamiltonian of a gapped Kitaev chain, src/examples/iMPS_pfaffian.py:6-11."""
    M = t1 * np.ones(2 * L - 1, complex)
    M[1::2] = t2
    M = np.diag(M, 1)
    return M + M.T.conj()


def test_pfaffian_H_to_iMPS():
    """pfaffian.H_to_iMPS on the example of src/examples/iMPS_pfaffian.py (parity-conserving input, one site per
    unit cell): oracle parity on the same finite MPS and the example's reconstruction check."""
    from temfpy_amd import gutzwiller, iMPS, pfaffian

    L, cut, chi, n_cell = 20, 10, 32, 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, err = pfaffian.H_to_iMPS(kitaev(L), kitaev(L + 1), {"chi_max": chi}, 1, cut, basis="M")
        ms = pfaffian.H_to_MPS(kitaev(L), {"chi_max": chi}, basis="M", ortho_center=cut)
        ml = pfaffian.H_to_MPS(kitaev(L + 1), {"chi_max": chi}, basis="M", ortho_center=cut)
        mv = pfaffian.H_to_MPS(kitaev(L + n_cell), {"chi_max": chi}, basis="M", ortho_center=cut)
    assert res.L == 1 and err.total_error < 1e-4 and set(np.unique(res.charges[0])) <= {0, 1}

    def sorted_dense(m):      # the package orders the indices of every bond by parity (stable)
        T = m.dense_tensors()
        q = gutzwiller.infer_parities(T)
        pm = [np.argsort(x, kind="stable") for x in q]
        return ([t[:, pm[i]][:, :, pm[i + 1]] for i, t in enumerate(T)], [np.asarray(x)[pm[b]] for b, x in enumerate(m.lam)],
                list(m.form))

    Ts, ls, fs = sorted_dense(ms)
    Tl, ll, fl = sorted_dense(ml)
    B, S, eo = io.mps_to_imps(Ts, ls, fs, Tl, ll, fl, 1, cut)
    # C_to_iMPS reports the left-hand errors only (pfaffian.py:2090); the unit cell with both rotations is MPS_to_iMPS
    np.testing.assert_allclose(list(err)[:2], eo[:2], rtol=0, atol=1e-9)
    assert err.right_unitary == 0.0 and err.right_schmidt == 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rot, err_rot = iMPS.MPS_to_iMPS(ms, ml, 1, cut, offset=0)
    np.testing.assert_allclose(list(err_rot), eo, rtol=0, atol=1e-9)
    for t, r, sl in zip(rot.dense_tensors(), B, S[:-1]):
        assert (sl[None, :, None] * np.abs(t - r)).max() < 1e-9
    # projected on the short chain's right Schmidt vectors instead of rotated: the same tensor up to the reported errors
    for t, r, sl in zip(res.dense_tensors(), B, S[:-1]):
        assert (sl[None, :, None] * np.abs(t - r)).max() < 10 * max(err_rot.total_error, 1e-9)
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, cut, n_cell)
    Tv, lv, fv = sorted_dense(mv)
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr, nv = io.overlap(Tr, lr, fr, Tr, lr, fr).real, io.overlap(Tv, lv, fv, Tv, lv, fv).real
    assert abs(abs(ov) / np.sqrt(nr * nv) - 1) < 1e-8


def test_gutzwiller_projected_chains_to_imps():
    """iMPS of a Gutzwiller-projected state from the projected finite chains of two lengths (``SpinMPSData`` input of
    ``iMPS.MPS_to_iMPS``, 2 S^z charge blocks): oracle parity on the same finite spin MPS and the reconstruction check
    of the reference's example (src/examples/iMPS.py:27-38) with projected chains."""
    from temfpy_amd import gutzwiller, iMPS, slater

    L, cut, chi, n_cell = 12, 6, 96, 2

    def spin_chain(n):
        C, _ = slater.correlation_matrix(ssh(n))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return gutzwiller.abrikosov_ph(slater.C_to_MPS(C, {"chi_max": chi}, spinful="PH", as_tenpy=False))

    ms, ml, mv = spin_chain(L), spin_chain(L + 2), spin_chain(L + 2 * n_cell)
    assert ms.conserve == "Sz" and ms.L == L
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, err = iMPS.MPS_to_iMPS(ms, ml, 2, cut)
    assert res.L == 2 and err.total_error < 1e-3
    Ts, ls, fs = dense(ms)
    Tl, ll, fl = dense(ml)
    B, S, eo = io.mps_to_imps(Ts, ls, fs, Tl, ll, fl, 2, cut)
    # (the Schmidt spectra of the projected chains reach down to the 1e-12 cutoff: the Procrustes rotations in directions
    # of weight < 1e-6 are fixed by rounding only, which the Schmidt-mixing measure and the weighted tensors see at 1e-7)
    np.testing.assert_allclose(list(err), eo, rtol=0, atol=1e-7)
    for a, b in zip(res.lam, S):
        np.testing.assert_allclose(np.sort(a)[::-1], np.sort(b)[::-1], rtol=0, atol=1e-12)
    # gauge-invariant comparison of the unit cells (SU(2) multiplets make the Schmidt spectrum of a projected chain
    # degenerate across and inside the 2 S^z sectors: the basis inside a multiplet is fixed by rounding, element-wise
    # parity does not exist): transfer matrix of the HIP cell against the oracle's, dominant eigenvalue 1, sub-leading
    # eigenvalues equal within the reported conversion error
    # (restricted to Schmidt states of weight > 1e-6: the rotation of the weaker ones - their weights go down to the 1e-12
    # cutoff of the projection - is fixed by rounding alone, here as in LAPACK, and a transfer matrix counts every state of
    # the bond alike, whatever its weight)
    def transfer(cell, lam):
        keep = [np.nonzero(np.asarray(x) > 1e-6)[0] for x in lam]
        E = None
        for j, t in enumerate(cell):
            t = t[:, keep[j]][:, :, keep[j + 1]]
            e = np.einsum("pab,pcd->acbd", t, np.conj(t)).reshape(t.shape[1] ** 2, t.shape[2] ** 2)
            E = e if E is None else E @ e
        return np.sort(np.abs(np.linalg.eigvals(E)))[::-1]
    th, to = transfer(res.dense_tensors(), res.lam), transfer(B, S)
    assert abs(th[0] - 1) < 1e-8 and np.abs(th[:6] - to[:6]).max() < 10 * err.total_error     # both cells carry that error
    for t, sl in zip(res.dense_tensors(), res.lam):     # right-canonical where it carries weight (the gauge blocks are rectangular
        X = np.einsum("pab,pcb->ac", t, t.conj())     #  where the two chains keep different numbers of states in a sector)
        k = np.nonzero(np.asarray(sl) > 1e-6)[0]
        assert np.abs(X[np.ix_(k, k)] - np.eye(len(k))).max() < 1e-4
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, cut, n_cell)
    Tv, lv, fv = dense(mv)
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr, nv = io.overlap(Tr, lr, fr, Tr, lr, fr).real, io.overlap(Tv, lv, fv, Tv, lv, fv).real
    assert abs(abs(ov) / np.sqrt(nr * nv) - 1) < 1e-5
    with pytest.raises(ValueError, match="FermionSite must conserve"):      # a spin cell is not a fermionic MPS (gutzwiller.py:172-176)
        gutzwiller.abrikosov_ph(res)
