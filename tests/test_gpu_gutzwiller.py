"""GPU parity tests of the Gutzwiller projections (temfpy_amd/gutzwiller.py, HIP path through the C ABI)
against the oracle (oracle/gutzwiller_oracle.py, itself pinned by brute force in test_oracle_gutzwiller.py).

Tolerances (fp64): Schmidt values 1e-10 abs (per bond, sorted), entropies 1e-10, norm of the projected state
1e-10 rel, state overlap 1 - |<oracle|hip>| <= 1e-10, right-canonical isometry 1e-10, kept bond dimensions
equal wherever no Schmidt value lies within a factor 10 of the cutoff."""
import os
import warnings

import numpy as np
import pytest

from oracle import gutzwiller_oracle as gw
from oracle import slater_oracle as orc
from tests_inputs import random_hopping, uniform_chain

pytestmark = pytest.mark.gpu
TOL_S = 1e-12
torch = pytest.importorskip("torch")


def hip_mps(H, chi, spinful, oc=None):
    from temfpy_amd import slater

    C, _ = slater.correlation_matrix(H)
    return slater.C_to_MPS(C, {"chi_max": chi}, spinful=spinful, ortho_center=oc, as_tenpy=False)


def oracle_inputs(mps):
    """Dense tensors, charges and centre Schmidt values of the HIP fermion MPS itself: the oracle projects
    the SAME fermion state, so that the comparison isolates the projection + canonicalisation."""
    T = mps.dense_tensors()
    q = [np.asarray(b.q_left) for b in mps.bonds]
    return T, q, mps.lam[mps.ortho_center], mps.ortho_center


def spin_overlap(B1, B2):
    E = np.ones((1, 1), complex)
    for a, b in zip(B1, B2):
        E = np.einsum("ab,pac,pbd->cd", E, a.conj(), b)
    return abs(E[0, 0])


def check(res, T, q, lam, oc, kind, cutoff=1e-12, isometry=1e-10):
    M, keep = gw.group_and_project(T, q, lam, oc, kind)
    B, S, nrm = gw.canonical_form_finite(M, cutoff)
    assert abs(res.norm / nrm - 1) < 1e-10
    assert res.L == len(M) and res.form == ["B"] * res.L
    for b, (a, r) in enumerate(zip(res.lam, S)):
        a, r = np.sort(a)[::-1], np.sort(r)[::-1]
        n = min(len(a), len(r))
        assert np.abs(a[:n] - r[:n]).max() < TOL_S, (b, np.abs(a[:n] - r[:n]).max())
        edge = (r < 10 * cutoff).sum() + (a < 10 * cutoff).sum()
        assert abs(len(a) - len(r)) <= edge, (b, len(a), len(r))
    Bh = res.dense_tensors()
    for t, sl in zip(Bh, res.lam):
        X = np.einsum("pab,pcb->ac", t, t.conj()) - np.eye(t.shape[1])
        if isometry is None:      # method="parallel": exact up to (cutoff / s)^2 per Schmidt index
            bound = 1e-10 + 4.0 * len(sl) * (cutoff / sl) ** 2
            assert np.all(np.abs(X) <= np.sqrt(np.outer(bound, bound))), np.abs(X).max()
        else:
            assert np.abs(X).max() < isometry
    assert abs(spin_overlap(B, Bh) - 1) < 1e-10
    return S


@pytest.mark.parametrize("L,seed,chi,oc", [(6, 1, 4096, None), (8, 2, 64, None), (10, 3, 48, 7), (16, 0, 64, None)])
def test_abrikosov_ph_random_hopping(L, seed, chi, oc):
    from temfpy_amd import gutzwiller

    H = random_hopping(L, seed)
    mps = hip_mps(H, chi, "PH", oc)
    res = gutzwiller.abrikosov_ph(mps)
    assert res.conserve == "Sz"
    S = check(res, *oracle_inputs(mps), "ph")
    # 2 S^z labels: steps of +-1 per site, zero at both ends
    assert res.charges[0].tolist() == [0] and res.charges[-1].tolist() == [0]
    for j, bl in enumerate(res.blocks):
        for p, ql, qr, *_ in bl:
            assert qr - ql == (1 if p == 1 else -1)
    np.testing.assert_allclose(res.entanglement_entropy(True),
                               [-(s**2 * np.log(s**2)).sum() for s in S], atol=1e-10)


@pytest.mark.parametrize("L,real", [(8, True), (12, True), (10, False)])
def test_abrikosov_ph_chains(L, real):
    from temfpy_amd import gutzwiller

    H = uniform_chain(L) + 0.05 * np.diag(np.cos(np.arange(L)))
    if not real:
        H = H + 0.1 * random_hopping(L, 5)
    mps = hip_mps(H, 96, "PH")
    res = gutzwiller.abrikosov_ph(mps)
    check(res, *oracle_inputs(mps), "ph")


@pytest.mark.parametrize("L,seed,chi", [(6, 2, 4096), (10, 4, 64), (16, 1, 64)])
def test_abrikosov_random_hopping(L, seed, chi):
    """Half filling is needed (total charge = number of spin sites): N = L spinful fermions on 2 L modes."""
    from temfpy_amd import gutzwiller, slater

    H = random_hopping(L, seed)
    C, _ = slater.correlation_matrix(H, N=L // 2)
    mps = slater.C_to_MPS(C, {"chi_max": chi}, spinful="simple", as_tenpy=False)
    res = gutzwiller.abrikosov(mps)
    assert res.conserve is None
    check(res, *oracle_inputs(mps), "std")


def test_reference_error_behaviour():
    from temfpy_amd import gutzwiller, slater

    C, _ = slater.correlation_matrix(random_hopping(5, 0), N=2)
    mps = slater.C_to_MPS(C, {"chi_max": 16}, as_tenpy=False)
    with pytest.raises(AssertionError, match="Odd-length"):
        gutzwiller.abrikosov_ph(mps)
    mps = slater.C_to_MPS(C, {"chi_max": 16}, spinful="simple", as_tenpy=False)   # N_total = 4 != 5
    with pytest.raises(AssertionError, match="Total charge must match"):
        gutzwiller.abrikosov(mps)
    C3, _ = slater.correlation_matrix(random_hopping(4, 0), N=1)
    m3 = slater.C_to_MPS(np.kron(C3, np.diag([1.0, 0.0])) , {"chi_max": 16}, as_tenpy=False)     # odd total charge
    with pytest.raises(AssertionError, match="parity of MPS must be even"):
        gutzwiller.abrikosov_ph(m3)
    mps = hip_mps(random_hopping(6, 1), 32, "PH")
    with pytest.raises(ValueError, match="does not divide"):
        gutzwiller.abrikosov_ph(mps, unit_cell_width=4)
    with pytest.warns(UserWarning, match="ignoring parity"):
        gutzwiller.abrikosov_ph(mps, parity=1)
    with pytest.warns(UserWarning, match="not in canonical form"):
        raw = gutzwiller.abrikosov_ph(mps, return_canonical=False)
    assert raw.form == [None] * raw.L
    T, q, lam, oc = oracle_inputs(mps)
    M, _ = gw.group_and_project(T, q, lam, oc, "ph")
    for a, b in zip(raw.dense_tensors(), M):
        np.testing.assert_allclose(a, b, atol=1e-12)
    # inplace
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = gutzwiller.abrikosov_ph(mps, inplace=True)
    assert out is None and isinstance(mps, gutzwiller.SpinMPSData) and mps.L == 6


def test_larger_chain_block_jacobi_and_cutoff():
    """L = 64 spinful chain at chi = 256: charge sectors beyond the LDS Jacobi, many Schmidt values below the
    cutoff (zeroed on the device, compacted on the host)."""
    from temfpy_amd import gutzwiller

    mps = hip_mps(uniform_chain(32) + 0.1 * np.diag(np.sin(np.arange(32.0))), 256, "PH")
    res = gutzwiller.abrikosov_ph(mps)
    check(res, *oracle_inputs(mps), "ph")


def test_exactly_symmetric_chain_rank_deficient_blocks():
    """Uniform chain (exact SU(2) and reflection symmetry): after the projection every charge block is exactly
    rank deficient with graded columns - the case in which a Gram-Schmidt QR normalised rounding noise
    (isometry off by 0.3) and which the Householder kernel handles without a rank decision."""
    from temfpy_amd import gutzwiller

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # the chi-truncated fermion MPS triggers the reference's diag_tol warning
        mps = hip_mps(uniform_chain(48), 128, "PH")
    res = gutzwiller.abrikosov_ph(mps)
    check(res, *oracle_inputs(mps), "ph")
    # S^z -> -S^z symmetry of the Schmidt spectrum at the centre
    b = res.L // 2
    q, lam = res.charges[b], res.lam[b]
    for c in np.unique(q[q > 0]):
        a, r = np.sort(lam[q == c])[::-1], np.sort(lam[q == -c])[::-1]
        n = min(len(a), len(r))
        assert np.abs(a[:n] - r[:n]).max() < 1e-3       # limited by the asymmetry of the chi-truncated INPUT (measured 3e-5)


@pytest.mark.parametrize("L,perturb", [(16, 0.3), (32, 0.1), (48, 0.0)])
def test_methods_agree(L, perturb):
    """method="parallel" (default: QR-only sweeps on two streams, all SVDs in one launch, one batched Gram-Schmidt pass
    over the kept rows) and "sequential" (TeNPy's algorithm step by step) against the oracle: same state, norm and
    Schmidt values, tensors right-isometric to 1e-10 in both."""
    from temfpy_amd import gutzwiller

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mps = hip_mps(uniform_chain(L) + perturb * np.diag(np.sin(np.arange(float(L)))), 128, "PH")
    seq = gutzwiller.abrikosov_ph(mps, method="sequential")
    check(seq, *oracle_inputs(mps), "ph")
    res = gutzwiller.abrikosov_ph(mps, method="parallel")
    check(res, *oracle_inputs(mps), "ph")
    assert abs(res.norm / seq.norm - 1) < 1e-12
    for a, b in zip(res.lam, seq.lam):
        a, b = np.sort(a)[::-1], np.sort(b)[::-1]
        n = min(len(a), len(b))
        assert np.abs(a[:n] - b[:n]).max() < 1e-12
        assert abs(len(a) - len(b)) <= (a < 1e-11).sum() + (b < 1e-11).sum()
    with pytest.raises(ValueError, match="method"):
        gutzwiller.abrikosov_ph(mps, method="fast")


@pytest.mark.parametrize("seed,kind", [(0, "ph"), (3, "ph"), (1, "std"), (2, "std")])
def test_pfaffian_input_parity_conserving(seed, kind):
    """Input from pfaffian.C_to_MPS (conserve = 'parity', complex tensors): charges are read off the tensors
    (infer_parities), no charge survives the projection (gutzwiller.py:244 / :443-444)."""
    from temfpy_amd import gutzwiller, pfaffian

    L = 8
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(2 * L), np.arange(2 * L), indexing="ij")
    M = rng.normal(size=(2 * L, 2 * L)) * np.exp(-abs(x - y) / 3.0)
    H = 1j * (M - M.T)                       # src/examples/pfaffian.py:13-17
    C = pfaffian.correlation_matrix(H, basis="M->M")
    C = C[0] if isinstance(C, tuple) else C
    mps = pfaffian.C_to_MPS(C, {"chi_max": 64}, basis="M")
    T = mps.dense_tensors()
    q = gutzwiller.infer_parities(T)
    total, fn = int(q[-1][0]), (gutzwiller.abrikosov_ph if kind == "ph" else gutzwiller.abrikosov)
    ok = total % 2 == 0 if kind == "ph" else total % 2 == (L // 2) % 2
    if not ok:
        with pytest.raises(AssertionError):
            fn(mps)
        return
    res = fn(mps)
    assert res.conserve is None
    M_, keep = gw.group_and_project(T, q, mps.lam[mps.ortho_center], mps.ortho_center, kind, conserve="parity")
    B, S, nrm = gw.canonical_form_finite(M_)
    assert abs(res.norm / nrm - 1) < 1e-10
    for b, (a, r) in enumerate(zip(res.lam, S)):
        a, r = np.sort(a)[::-1], np.sort(r)[::-1]
        n = min(len(a), len(r))
        assert np.abs(a[:n] - r[:n]).max() < TOL_S and abs(len(a) - len(r)) <= (r < 1e-11).sum() + (a < 1e-11).sum()
    Bh = res.dense_tensors()
    for t in Bh:
        X = np.einsum("pab,pcb->ac", t, t.conj())
        assert np.abs(X - np.eye(len(X))).max() < 1e-10
    assert abs(spin_overlap(B, Bh) - 1) < 1e-10


def test_config5_full_size_properties():
    """BASELINE config 5 at full size (L = 512 uniform chain, spinful "PH", chi_max = 512 -> 512 spins), checked
    through size-independent properties: right-canonical isometry on sampled sites, <psi|psi> = 1 by transfer
    matrices, 2 S^z bookkeeping, and agreement of the two canonicalisation methods (norm, Schmidt values)."""
    from temfpy_amd import gutzwiller, slater

    C, _ = slater.correlation_matrix(uniform_chain(512))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mps = slater.C_to_MPS(C, {"chi_max": 512}, spinful="PH", as_tenpy=False)
    seq = gutzwiller.abrikosov_ph(mps)
    par = gutzwiller.abrikosov_ph(mps, method="parallel")
    assert seq.L == 512 and seq.conserve == "Sz" and 1e-60 < seq.norm < 1e-50
    assert abs(par.norm / seq.norm - 1) < 1e-10
    T = seq.dense_tensors()
    for j in (0, 1, 100, 255, 256, 400, 510, 511):
        X = np.einsum("pab,pcb->ac", T[j], T[j].conj())
        assert np.abs(X - np.eye(len(X))).max() < 1e-10
    E = np.ones((1, 1))
    for t in T:
        E = np.tensordot(np.tensordot(E, t.conj(), axes=(0, 1)), t, axes=([0, 1], [1, 0]))
    assert abs(E[0, 0] - 1) < 1e-9
    assert seq.charges[0].tolist() == [0] and seq.charges[-1].tolist() == [0]
    for bl in (seq.blocks[7], seq.blocks[300]):
        for p, ql, qr, *_ in bl:
            assert qr - ql == (1 if p == 1 else -1)
    for b in range(0, 513, 16):
        a, r = np.sort(seq.lam[b])[::-1], np.sort(par.lam[b])[::-1]
        n = min(len(a), len(r))
        assert np.abs(a[:n] - r[:n]).max() < 1e-11 and abs(len(a) - len(r)) <= (a < 1e-11).sum() + (r < 1e-11).sum()
    S = seq.entanglement_entropy()
    assert abs(S[255] - S[255 - 2]) < 0.5 and 0.5 < S[255] < 2.0          # log-law plateau of the projected chain


def test_rescaling_inside_the_sweeps_leaves_the_result_alone_and_lifts_the_underflow():
    """The triangular factors of the two QR sweeps are rescaled by a common power of two after every step
    (tmf_rescale_pow2_batched; TeNPy's canonical_form_finite renormalises every step): powers of two are exact, so the
    result of a short chain is THE SAME with and without (Schmidt values, tensors up to rounding of the different scale of
    intermediate products: 1e-13), and a chain whose projected norm leaves the range of a double (0.6 per spin: 2 200 spins)
    is projected like any other - `norm` then underflows and `log2_norm` carries it."""
    from temfpy_amd import gutzwiller, slater

    # (a) same result as without rescaling (the switch is read per call)
    C, _ = slater.correlation_matrix(uniform_chain(24))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mps = slater.C_to_MPS(C, {"chi_max": 64}, spinful="PH", as_tenpy=False)
        on = gutzwiller.abrikosov_ph(mps)
        os.environ["TMF_GW_RESCALE"] = "0"
        try:
            off = gutzwiller.abrikosov_ph(mps)
        finally:
            del os.environ["TMF_GW_RESCALE"]
    assert abs(on.norm / off.norm - 1) < 1e-13 and abs(on.log2_norm - np.log2(off.norm)) < 1e-10
    for a, b in zip(on.lam, off.lam):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)
    assert abs(spin_overlap(on.dense_tensors(), off.dense_tensors()) - 1) < 1e-12
    # (b) long chains against the oracle run with a renormalisation per step of its first sweep.  (Without the rescaling
    # the SQUARES of the triangular factors leave the range at ~1000 spins and the result is silently wrong.)
    def stable(M):
        M = [np.array(m, complex) for m in M]
        lg = 0.0
        for j in range(len(M) - 1):
            d, cl, cr = M[j].shape
            Q, R = np.linalg.qr(M[j].reshape(d * cl, cr))
            sc = np.abs(R).max()
            lg += np.log2(sc)
            M[j] = Q.reshape(d, cl, -1)
            M[j + 1] = np.einsum("ab,pbc->pac", R / sc, M[j + 1])
        _, S, nrm = gw.canonical_form_finite(M, 1e-12)
        return S, lg + np.log2(nrm)

    for Ls in (1200, 2200):
        C, _ = slater.correlation_matrix(uniform_chain(Ls))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mps = slater.C_to_MPS(C, {"chi_max": 16}, spinful="PH", as_tenpy=False)
            results = [gutzwiller.abrikosov_ph(mps), gutzwiller.abrikosov_ph(mps, method="sequential")]
        T, q, lam, oc = oracle_inputs(mps)
        M, _ = gw.group_and_project(T, q, lam, oc, "ph")
        S, lg = stable(M)
        for res in results:
            assert res.L == Ls and abs(res.log2_norm - lg) < 1e-9 and lg < -500.0
            assert abs(res.norm / 2.0 ** lg - 1) < 1e-9 if Ls == 1200 else res.norm < 1e-300
            for x, r in zip(res.lam, S):
                x, r = np.sort(x)[::-1], np.sort(r)[::-1]
                n = min(len(x), len(r))
                assert np.abs(x[:n] - r[:n]).max() < 1e-12
            for t in res.dense_tensors()[Ls // 2 - 2: Ls // 2 + 2]:
                X = np.einsum("pab,pcb->ac", t, t.conj()) - np.eye(t.shape[1])
                assert np.abs(X).max() < 1e-9


def test_tables_built_while_the_sweeps_run_give_the_same_result():
    """From 96 spins on the parallel method starts its two sweeps when the tables of their first 16 steps are up and builds
    the rest (and the all-bond tail) under them: same launches, same arithmetic - the result must equal the one-piece form
    (TMF_GW_CHUNKS=0) bit for bit."""
    from temfpy_amd import gutzwiller, slater

    for Ls, real in ((130, True), (100, False)):
        H = uniform_chain(Ls) if real else random_hopping(Ls, 5)
        C, _ = slater.correlation_matrix(H, Ls // 2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mps = slater.C_to_MPS(C, {"chi_max": 48}, spinful="PH", as_tenpy=False)
            a = gutzwiller.abrikosov_ph(mps)
            os.environ["TMF_GW_CHUNKS"] = "0"
            try:
                b = gutzwiller.abrikosov_ph(mps)
            finally:
                del os.environ["TMF_GW_CHUNKS"]
        assert a.norm == b.norm and a.log2_norm == b.log2_norm
        for x, y in zip(a.lam, b.lam):
            assert np.array_equal(x, y)
        for x, y in zip(a.dense_tensors(), b.dense_tensors()):
            assert np.array_equal(x, y)


def test_adapter_from_flat_tables_equals_the_per_site_adapter():
    """gutzwiller._fermions_from_shard (all block records of a conversion at once from its flat tables) against the loop over
    the per-site objects: the same records, offsets into the same page-locked buffer and charges."""
    from temfpy_amd import gutzwiller, slater

    for spinful, L in (("PH", 24), ("simple", 10), (None, 20)):
        C, _ = slater.correlation_matrix(uniform_chain(L))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mps = slater.C_to_MPS(C, {"chi_max": 64}, spinful=spinful, as_tenpy=False, ortho_center=None if spinful else 7)
        a, b = gutzwiller._fermions_from_shard(mps), gutzwiller._fermions_from_sites(mps)
        assert a is not None and len(a.blocks) == len(b.blocks) > 0
        key = lambda r: tuple(int(x) for x in r)      # noqa: E731
        assert sorted(map(key, a.blocks)) == sorted(map(key, b.blocks))
        assert [key(r) for r in a.blocks] == [key(r) for r in b.blocks]      # and in the same order
        assert all(np.array_equal(x, y) for x, y in zip(a.charges, b.charges))
        assert a.flat is b.flat and a.dtype == b.dtype and np.array_equal(a.lam_c, b.lam_c)


def test_chi512_sample_against_oracle():
    """Config-5 bond dimension (chi_max = 512, 140-state charge sectors, the slab QR + preconditioned Jacobi
    path at full block size) on a 64-spin chain against the charge-block oracle: Schmidt values 1e-12 (default method:
    3e-12 for the values below 3e-11, see the comment)."""
    from temfpy_amd import gutzwiller, slater

    C, _ = slater.correlation_matrix(uniform_chain(64))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mps = slater.C_to_MPS(C, {"chi_max": 512}, spinful="PH", as_tenpy=False)
    T, q, lam, oc = oracle_inputs(mps)
    M, keep = gw.group_and_project(T, q, lam, oc, "ph")
    B, S, Q, nrm = gw.canonical_form_finite_blocks(M, gw.spin_charges(q, keep))
    for method in ("parallel", "sequential"):
        res = gutzwiller.abrikosov_ph(mps, method=method)
        assert abs(res.norm / nrm - 1) < 1e-10
        for b, (a, r) in enumerate(zip(res.lam, S)):
            a, r = np.sort(a)[::-1], np.sort(r)[::-1]
            n = min(len(a), len(r))
            # "parallel" truncates every bond on its own: values within a decade of the cutoff (1e-12) keep the weight
            # that the step-by-step algorithm has already truncated away to the right of the site - up to ~cutoff more
            tol = TOL_S if method == "sequential" else np.where(r[:n] < 3e-11, 3e-12, TOL_S)
            assert np.all(np.abs(a[:n] - r[:n]) < tol), (method, b, np.abs(a[:n] - r[:n]).max())
            assert abs(len(a) - len(r)) <= (a < 1e-11).sum() + (r < 1e-11).sum()
    for b in (1, 16, 32, 63):      # 2 S^z labels of the kept Schmidt indices: same multiset per charge
        for c in np.unique(Q[b]):
            assert abs((res.charges[b] == c).sum() - (Q[b] == c).sum()) <= 2


# ---------------------------------------------------------------------------------------------------
# infinite MPS (gutzwiller.py:197-206, :272 / :475)
# ---------------------------------------------------------------------------------------------------
def _ssh(L, t1=-1.5, t2=-1.0):
    M = t1 * np.ones(L - 1)
    M[1::2] = t2
    M = np.diag(M, 1)
    return M + M.T


def _mixed_transfer_dominant(A, B):
    E = None
    for a, b in zip(A, B):
        e = np.einsum("pab,pcd->acbd", a, np.conj(b)).reshape(a.shape[1] * b.shape[1], a.shape[2] * b.shape[2])
        E = e if E is None else E @ e
    return np.abs(np.linalg.eigvals(E)).max()


@pytest.mark.parametrize("kind,spinful,chi", [("ph", "PH", 24), ("std", "simple", 24), ("ph", "PH", 40)])
def test_infinite_mps_against_oracle(kind, spinful, chi):
    """Projection of the unit cell of an infinite fermionic MPS (``slater.H_to_iMPS``) against the dense restatement
    (oracle ``group_and_project_cell`` + ``canonical_form_infinite``) fed with the SAME cell: Schmidt values of every bond
    (1e-8: both sides take them from Gram matrices, i.e. with sqrt(eps) resolution), the state per unit cell (dominant
    eigenvalue of the mixed transfer matrix = 1 within 1e-8), norm per cell, right-canonical tensors (1e-10), left
    environment diag(lam^2) (1e-7), 2 S^z labels."""
    from temfpy_amd import gutzwiller, slater

    L, cut = 24, 12
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cell, _ = slater.H_to_iMPS(_ssh(L), _ssh(L + 2), {"chi_max": chi}, 2, cut, spinful=spinful)
        assert cell.L == 4 and cell.conserve == "N"
        if kind == "ph":
            out = gutzwiller.abrikosov_ph(cell)
        else:
            with pytest.raises(ValueError, match="q_left"):
                gutzwiller.abrikosov(cell)
            with pytest.raises(ValueError, match="charge sector"):
                gutzwiller.abrikosov(cell, q_left=10**6)
            out = gutzwiller.abrikosov(cell, q_left=0)
    assert out.bc == "infinite" and out.L == 2 and out.form == ["B", "B"]
    T, q = cell.dense_tensors(), [np.asarray(x) for x in cell.charges[:4]]
    M, keep = gw.group_and_project_cell(T, q, cell.cell_charge, kind, "N", 0, 0)
    Bo, So, eta = gw.canonical_form_infinite(M)
    assert abs(out.norm - np.sqrt(eta)) < 1e-9 * np.sqrt(eta)
    Bd = out.dense_tensors()
    for b in range(3):
        a, r = np.sort(out.lam[b])[::-1], np.sort(So[b])[::-1]
        n = min(len(a), len(r))
        assert n > 0 and np.abs(a[:n] - r[:n]).max() < 1e-8, (b, len(a), len(r))
        assert np.all(a[n:] < 1e-6) and np.all(r[n:] < 1e-6)
    np.testing.assert_array_equal(out.lam[0], out.lam[2])
    np.testing.assert_array_equal(out.charges[0], out.charges[2])
    for j, t in enumerate(Bd):
        X = np.einsum("pab,pcb->ac", t, t.conj())
        assert np.abs(X - np.eye(len(X))).max() < 1e-10
        l2 = sum(t[p].conj().T @ np.diag(out.lam[j] ** 2) @ t[p] for p in range(2))
        assert np.abs(l2 - np.diag(out.lam[j + 1] ** 2)).max() < 1e-7
    # the same state per unit cell as the oracle's, and as the projected (non-canonical) cell itself
    own = _mixed_transfer_dominant(Bd, Bd)
    assert abs(own - 1) < 1e-9
    assert abs(_mixed_transfer_dominant(Bd, Bo) / np.sqrt(own * _mixed_transfer_dominant(Bo, Bo)) - 1) < 1e-8
    assert abs(_mixed_transfer_dominant(Bd, M) / np.sqrt(own * eta) - 1) < 1e-8
    if kind == "ph":       # 2 S^z labels: p = 0 lowers, p = 1 raises; the last site closes the cell with cell_charge
        assert out.conserve == "Sz" and out.cell_charge == cell.cell_charge - 2
        for j, bl in enumerate(out.blocks):
            for p, ql, qr, l0, l1, r0, r1, a in bl:
                assert ql + (1 if p else -1) == qr + (out.cell_charge if j == out.L - 1 else 0)
                assert np.all(out.charges[j][l0:l1] == ql) and np.all(out.charges[j + 1][r0:r1] == qr)
    else:
        assert out.conserve is None
    # not canonical: the projected cell as it is
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        raw = gutzwiller.abrikosov_ph(cell, return_canonical=False) if kind == "ph" else gutzwiller.abrikosov(
            cell, q_left=0, return_canonical=False)
    assert raw.form == [None, None]
    assert abs(_mixed_transfer_dominant(raw.dense_tensors(), M) / eta - 1) < 1e-10
    if kind == "ph":       # offset shifts every 2 S^z label (gutzwiller.py:333); inplace turns the cell itself into the result
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            shifted = gutzwiller.abrikosov_ph(cell, offset=3)
            assert gutzwiller.abrikosov_ph(cell, inplace=True) is None
        for a, b in zip(shifted.charges, out.charges):
            np.testing.assert_array_equal(a, b - 3)
        assert shifted.cell_charge == out.cell_charge
        assert isinstance(cell, gutzwiller.SpiniMPSData) and cell.bc == "infinite" and cell.L == 2
        for a, b in zip(cell.lam, out.lam):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


def test_infinite_mps_parity_conserving_complex():
    """Unit cell of a Kitaev chain (``pfaffian.H_to_iMPS``: parity labels, complex tensors, two fermion sites = one spin site
    per cell): whichever projection the parity of the cell admits, against the dense restatement."""
    from temfpy_amd import gutzwiller, pfaffian

    t1, t2, L, cut = 1.5j, 1j, 20, 10

    def kitaev(n):
        M = t1 * np.ones(2 * n - 1, complex)
        M[1::2] = t2
        M = np.diag(M, 1)
        return M + M.T.conj()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cell, _ = pfaffian.H_to_iMPS(kitaev(L), kitaev(L + 2), {"chi_max": 16}, 2, cut, basis="M")
        assert cell.L == 2 and cell.conserve == "parity"
        if cell.cell_charge % 2 == 0:
            kind, out = "ph", gutzwiller.abrikosov_ph(cell, parity=1)
            with pytest.raises(AssertionError, match="Total charge"):
                gutzwiller.abrikosov(cell, q_left=0)
            par = 1
        else:
            kind, out = "std", gutzwiller.abrikosov(cell, q_left=1)
            with pytest.raises(AssertionError, match="parity"):
                gutzwiller.abrikosov_ph(cell)
            par = 0
    assert out.bc == "infinite" and out.L == 1 and out.conserve is None
    T, q = cell.dense_tensors(), [np.asarray(x) for x in cell.charges[:2]]
    M, keep = gw.group_and_project_cell(T, q, cell.cell_charge, kind, "parity", par, 1)
    Bo, So, eta = gw.canonical_form_infinite(M)
    assert abs(out.norm - np.sqrt(eta)) < 1e-9 * np.sqrt(eta)
    Bd = out.dense_tensors()
    a, r = np.sort(out.lam[0])[::-1], np.sort(So[0])[::-1]
    n = min(len(a), len(r))
    assert n > 0 and np.abs(a[:n] - r[:n]).max() < 1e-8 and np.all(a[n:] < 1e-6) and np.all(r[n:] < 1e-6)
    X = np.einsum("pab,pcb->ac", Bd[0], Bd[0].conj())
    assert np.abs(X - np.eye(len(X))).max() < 1e-10
    own = _mixed_transfer_dominant(Bd, Bd)
    assert abs(own - 1) < 1e-9
    assert abs(_mixed_transfer_dominant(Bd, Bo) / np.sqrt(own * _mixed_transfer_dominant(Bo, Bo)) - 1) < 1e-8
    assert abs(_mixed_transfer_dominant(Bd, M) / np.sqrt(own * eta) - 1) < 1e-8
    assert out.timings and gutzwiller.SpiniMPSData is type(out)


def test_tenpy_object_returned_by_the_package_is_accepted():
    """With TeNPy installed ``slater.C_to_MPS`` returns a TeNPy ``MPS`` (as the reference does); ``to_tenpy`` leaves a
    reference to the device-side container on it, through which the projections and ``iMPS.MPS_to_iMPS`` continue.  TeNPy
    is not installed here: a stand-in object carries the reference."""
    from temfpy_amd import gutzwiller, iMPS

    class Stand:
        def __init__(self, own):
            self._temfpy_amd, self.L = own, own.L

    mps = hip_mps(uniform_chain(8), 256, "PH")
    a = gutzwiller.abrikosov_ph(mps)
    b = gutzwiller.abrikosov_ph(Stand(mps))          # (no TeNPy: the native container comes back)
    for x, y in zip(a.lam, b.lam):
        np.testing.assert_array_equal(x, y)
    s = Stand(mps)
    assert gutzwiller.abrikosov_ph(s, inplace=True) is None and isinstance(s, gutzwiller.SpinMPSData)
    bad = Stand(mps)
    bad.L = 4
    with pytest.raises(ValueError, match="modified"):
        gutzwiller.abrikosov_ph(bad)
    assert gutzwiller.native(a) is a and iMPS.MPS_to_iMPS is not None


def _random_cell(rng, L, Q, conserve, cplx):
    """Random charge-conserving unit cell of L fermion sites with Q particles per cell (labels mod 2 for 'parity')."""
    from temfpy_amd.iMPS import iMPSData

    mod = 2 if conserve == "parity" else None
    if mod:
        q = [np.sort(np.concatenate((np.arange(2), rng.integers(0, 2, size=rng.integers(3, 7))))) for _ in range(L)]
    else:     # a window of four values that drifts with the filling
        q = [np.sort(np.concatenate((np.arange(4), rng.integers(0, 4, size=rng.integers(2, 6))))) + (i * Q) // L for i in range(L)]
    blocks = []
    for i in range(L):
        ql = q[i]
        qr = q[i + 1] if i + 1 < L else ((q[0] + Q) % 2 if mod else q[0] + Q)       # labels in the convention q_l + p = q_r
        bl = []
        for p in (0, 1):
            for cl in np.unique(ql):
                rows = np.nonzero(ql == cl)[0]
                cols = np.nonzero(qr == ((cl + p) % 2 if mod else cl + p))[0]
                if len(rows) and len(cols):
                    assert np.all(np.diff(cols) == 1) or mod      # (parity: the shifted labels of the last bond are not sorted)
                    a = rng.normal(size=(len(rows), len(cols))) + (1j * rng.normal(size=(len(rows), len(cols))) if cplx else 0)
                    lab = (cl + p - (Q if i + 1 == L else 0))
                    lab = int(lab % 2) if mod else int(lab)
                    if np.all(np.diff(cols) == 1):
                        bl.append((p, int(cl), lab, int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1, a))
                    else:
                        raise AssertionError("non-contiguous block")
        blocks.append(bl)
    lam = [np.ones(len(x)) / np.sqrt(len(x)) for x in q] + [np.ones(len(q[0])) / np.sqrt(len(q[0]))]
    return iMPSData(blocks, lam, q + [q[0]], (Q % 2 if mod else Q), max(L // 2, 1), conserve=conserve), q


@pytest.mark.parametrize("kind,seed,cplx,L,conserve", [("ph", 0, True, 6, "N"), ("std", 1, True, 6, "N"), ("ph", 2, False, 6, "N"),
                                                      ("std", 3, False, 6, "N"), ("ph", 4, True, 4, "parity"),
                                                      ("std", 5, True, 2, "parity"), ("std", 6, False, 6, "parity"),
                                                      ("ph", 7, False, 2, "N"), ("std", 8, True, 4, "N")])
def test_infinite_mps_hand_made_cells(kind, seed, cplx, L, conserve):
    """Random charge-conserving cells (one to three spin sites per cell, number or parity labels, several sectors per bond,
    some of them on no closed path through the cell): sector bookkeeping of the periodic projector against the dense
    restatement.  (The cell is not canonical and not normalised: the projection does not care.)"""
    from temfpy_amd import gutzwiller

    rng = np.random.default_rng(seed)
    if kind == "std":
        Q = L // 2                                     # one particle per spin site (gutzwiller.py:178-187)
    else:
        Q = 2 * (L // 4) + 2 if conserve == "N" else 0   # even (gutzwiller.py:374-376), about half filling
    cell, q = _random_cell(rng, L, Q, conserve, cplx)
    q_left = 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = gutzwiller.abrikosov_ph(cell) if kind == "ph" else gutzwiller.abrikosov(cell, q_left=q_left)
    M, keep = gw.group_and_project_cell(cell.dense_tensors(), q, Q, kind, conserve, 0, q_left)
    Bo, So, eta = gw.canonical_form_infinite(M)
    Ls = L // 2
    assert out.L == Ls and abs(out.norm - np.sqrt(eta)) < 1e-9 * np.sqrt(eta)
    Bd = out.dense_tensors()
    for b in range(Ls + 1):
        a, r = np.sort(out.lam[b])[::-1], np.sort(So[b])[::-1]
        n = min(len(a), len(r))
        assert n > 0 and np.abs(a[:n] - r[:n]).max() < 1e-8 and np.all(a[n:] < 5e-6) and np.all(r[n:] < 5e-6), b      # (states the Gram matrices resolve as noise: weight < 3e-11)
    for t in Bd:      # (1e-7: where a noise state was removed from a bond, the rows to its left lost that column and were rescaled)
        X = np.einsum("pab,pcb->ac", t, t.conj())
        assert np.abs(X - np.eye(len(X))).max() < 1e-7
    own = _mixed_transfer_dominant(Bd, Bd)
    assert abs(own - 1) < 1e-7
    assert abs(_mixed_transfer_dominant(Bd, Bo) / np.sqrt(own * _mixed_transfer_dominant(Bo, Bo)) - 1) < 1e-7
    assert abs(_mixed_transfer_dominant(Bd, M) / np.sqrt(own * eta) - 1) < 1e-7
