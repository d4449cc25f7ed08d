"""CPU tests of the C-ABI host entry points (tmf_cut_vectors, tmf_site_prepare) against the
oracle and the reference fixtures.  No GPU calls."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from oracle import slater_oracle as orc
from temfpy_amd import _native as nat


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def masks_to_bool(sets, k):
    out = np.zeros((len(sets), k), bool)
    for i in range(k):
        out[:, i] = (sets[:, i // 64] >> np.uint64(i % 64)) & np.uint64(1)
    return out


def test_library_exports_every_symbol():
    lib = nat.load()
    for s in nat.SYMBOLS:
        assert hasattr(lib, s)
    assert lib.tmf_version() >= 100
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "temfpy_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(tmf_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(nat.SYMBOLS), declared ^ set(nat.SYMBOLS)


@pytest.mark.parametrize("name", golden_names())
def test_cut_vectors_matches_reference_fixture(name):
    g = load(name)
    L = int(g["L"])
    for b in range(L + 1):
        e = g[f"b{b}_e"]
        sets, lam, q, _ = nat.cut_vectors(e, int(g[f"b{b}_nfilled"][0]), int(g["chi_max"]), 1e-6, 1e-12)
        ref = g[f"b{b}_sets"]
        got = masks_to_bool(sets, e.size)
        if not np.array_equal(got, ref):
            # only the order inside a numerically degenerate multiplet may differ (libm log vs numpy log)
            assert got.shape == ref.shape
            assert sorted(map(bytes, got)) == sorted(map(bytes, ref))
            np.testing.assert_allclose(np.sort(lam), np.sort(g[f"b{b}_lam_raw"]), rtol=1e-12)
        else:
            np.testing.assert_allclose(lam, g[f"b{b}_lam_raw"], rtol=1e-13, atol=0)
        # charges ascending, sector boundaries as in the fixture
        qs, start = np.unique(q, return_index=True)
        np.testing.assert_array_equal(qs, g[f"b{b}_q"])
        np.testing.assert_array_equal(start, g[f"b{b}_qstart"])


def test_cut_vectors_edge_cases():
    sets, lam, q, _ = nat.cut_vectors(np.zeros(0), 3, 8, 1e-6, 1e-12)
    assert sets.shape == (1, 2) and lam[0] == 1.0 and q[0] == 3
    sets, lam, q, _ = nat.cut_vectors(np.zeros(0), 3, 8, 1e-6, 1e-12, sectors=[2])
    assert len(lam) == 0
    # sector filter
    e = np.array([0.9, 0.5, 0.2])
    sets, lam, q, _ = nat.cut_vectors(e, 1, 0, 1e-6, 1e-12, sectors=[2])
    assert np.all(q == 2) and len(q) == 3
    # unlimited chi, all 8 subsets
    sets, lam, q, _ = nat.cut_vectors(e, 0, 0, 1e-6, 1e-12)
    assert len(lam) == 8 and abs((lam**2).sum() - 1) < 1e-14
    # degenerate pair straddling chi_max is dropped whole (schmidt_utils.py:163-185)
    e2 = 1 / (1 + np.exp(2 * np.array([1.0, 1.0, 3.0])))
    sets, lam, q, _ = nat.cut_vectors(e2, 0, 2, 1e-6, 1e-12)
    assert len(lam) == 1
    with pytest.raises(NotImplementedError):
        nat.cut_vectors(np.full(129, 0.5), 0, 4, 1e-6, 1e-12)


def _device_layout(cut, side):
    """V = [entangled | filled] as the HIP path stores it, from an oracle Cut."""
    if side == "L":
        v, nf, k = cut.vL, cut.nfL, cut.k
        return np.concatenate((v[:, nf:nf + k], v[:, :nf]), axis=1), nf
    v, nf, k = cut.vR, cut.nfR, cut.k
    n0 = cut.nR - nf - k
    return np.concatenate((v[:, n0:n0 + k], v[:, n0 + k:]), axis=1), nf


def _masks(cut):
    m = np.zeros((len(cut.sets), 2), np.uint64)
    for i in range(cut.k):
        m[:, i // 64] |= cut.sets[:, i].astype(np.uint64) << np.uint64(i % 64)
    return m


@pytest.mark.parametrize("name", ["rand_L16_s0_chi32", "rand_L24_s3_oc7_chi48", "chain_L16_chi32",
                                  "chainPH_L8_chi64", "rand_L8_s0_chi8", "randSimple_L6_s4_chi32"])
def test_site_prepare_matches_oracle(name):
    g = load(name)
    kw = {}
    if "kw_ortho_center" in g:
        kw["ortho_center"] = int(g["kw_ortho_center"])
    if "kw_spinful" in g:
        kw["spinful"] = str(g["kw_spinful"])
    cuts, sites = orc.c_to_mps(g["C_in"], {"chi_max": int(g["chi_max"])}, **kw)
    L, oc = int(g["L"]), int(g["ortho_center"])
    for i in range(L):
        s = sites[i]
        mode = 0 if s.mode == "left" else 1
        side = "L" if mode == 0 else "R"
        bra, ket = (cuts[i], cuts[i + 1]) if mode == 0 else (cuts[i + 1], cuts[i])
        Vb, nfb = _device_layout(bra, side)
        Vk, nfk = _device_layout(ket, side)
        qb = bra.n_filled("L") + bra.sets.sum(axis=1)
        qk = ket.n_filled("L") + ket.sets.sum(axis=1)
        r = nat.site_prepare(mode, bra.k, nfb, _masks(bra), qb, ket.k, nfk, _masks(ket), qk)
        np.testing.assert_array_equal(r["bra_p"], s.bra_p)
        np.testing.assert_array_equal(r["bra_alpha"], s.bra_alpha)
        assert (r["sb"], r["sk"]) == s.M.shape
        # W assembled from the selections reproduces det_always and the Schur complement
        nb = Vb.shape[0]
        Vk_sub = Vk[1:] if mode == 1 else Vk[:nb]
        phys = Vk[0] if mode == 1 else Vk[nb]
        Ofull = Vb.conj().T @ Vk_sub
        W = np.zeros((r["mb"], r["mk"]), complex)
        for a, (rs, sg) in enumerate(zip(r["row_sel"], r["row_sign"])):
            row = phys[r["col_sel"]] if rs < 0 else Ofull[rs, r["col_sel"]]
            W[a] = sg * row * r["col_sign"]
        k = r["k"]
        if k:
            det = np.linalg.det(W[:k, :k])
            M = W[k:, k:] - W[k:, :k] @ np.linalg.inv(W[:k, :k]) @ W[:k, k:]
        else:
            det, M = 1.0, W
        np.testing.assert_allclose(det, s.det_always, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(M, s.M, rtol=0, atol=1e-10)
        # sectors and index lists
        assert sorted(int(x) for x in r["sectors"]["q"]) == sorted(s.blocks)
        for sec in r["sectors"]:
            r0, r1, c0, c1, blk = s.blocks[int(sec["q"])]
            assert (sec["r0"], sec["r1"], sec["c0"], sec["c1"]) == (r0, r1, c0, c1)
            n = int(sec["n"])
            bl = r["idx_pool"][sec["bra_off"]: sec["bra_off"] + (r1 - r0) * n].reshape(r1 - r0, n)
            kl = r["idx_pool"][sec["ket_off"]: sec["ket_off"] + (c1 - c0) * n].reshape(c1 - c0, n)
            np.testing.assert_array_equal(bl, np.nonzero(s.sets_bra[r0:r1])[1].reshape(r1 - r0, n))
            np.testing.assert_array_equal(kl, np.nonzero(s.sets_ket[c0:c1])[1].reshape(c1 - c0, n))


def test_report_schmidt_checks_follows_test_action():
    """testing.report_schmidt_checks mirrors the reference's TEST_ACTION switch (testing.py:15-128)."""
    import warnings
    from temfpy_amd import testing

    old = testing.TEST_ACTION
    try:
        dev = {"vL does not diagonalise C_LL": 3e-7, "vL is not unitary": 1e-15}
        testing.TEST_ACTION = "warn"
        with pytest.warns(testing.ComparisonWarning, match="vL does not diagonalise C_LL"):
            testing.report_schmidt_checks(dev, 1e-8)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            testing.report_schmidt_checks(dev, 1e-6)           # within tolerance: silent
        testing.TEST_ACTION = "raise"
        with pytest.raises(AssertionError, match="3e-07"):
            testing.report_schmidt_checks(dev, 1e-8)
        with pytest.raises(AssertionError):
            testing.report_schmidt_checks({"x": float("nan")}, 1e-8)   # NaN never passes
        testing.TEST_ACTION = "pass"
        testing.report_schmidt_checks(dev, 1e-8)
        testing.TEST_ACTION = "bogus"
        with pytest.raises(ValueError, match="TEST_ACTION"):
            testing.report_schmidt_checks(dev, 1e-8)
    finally:
        testing.TEST_ACTION = old


def test_batch_error_message_reaches_the_calling_thread():
    """tmf_last_error is thread-local; a failure inside a worker thread of the batched host entry points must
    still surface with its text (limit: 128 entangled orbitals per cut) on the thread that called."""
    lib = nat.load()
    ncut = 12
    k = np.full(ncut, 4, np.int32)
    k[9] = 130                               # handled by a worker thread, not by the caller
    e_off = np.concatenate(([0], np.cumsum(k)[:-1])).astype(np.int64)
    e_pool = np.linspace(0.1, 0.9, int(k.sum()))
    nfl = np.zeros(ncut, np.int32)
    cap = 9
    sets, lam, q = np.zeros((ncut, cap, 2), np.uint64), np.zeros((ncut, cap)), np.zeros((ncut, cap), np.int32)
    chi, chk = np.zeros(ncut, np.int64), np.zeros(ncut, np.int64)
    st = lib.tmf_cut_vectors_batch(ncut, nat._p(e_pool), nat._p(e_off), nat._p(k), nat._p(nfl), 8, 1e-6, 1e-12, None, 0, cap,
                                   nat._p(sets), nat._p(lam), nat._p(q), nat._p(chi), nat._p(chk), 4)
    assert st == -3
    msg = lib.tmf_last_error().decode()
    assert "130 entangled orbitals" in msg and "item 9" in msg, msg
    with pytest.raises(NotImplementedError, match="130 entangled orbitals"):
        nat.check(st, "tmf_cut_vectors_batch")


@pytest.mark.skipif(not os.environ.get("TMF_ASAN_LIB"), reason="set TMF_ASAN_LIB to the sanitizer build (make -C temfpy_amd/csrc asan)")
def test_sanitizer_build_marker():
    """`make -C temfpy_amd/csrc asan` builds host_enum.cpp with -fsanitize=address,undefined into
    libtemfpy_host_asan.so; tools/run_host_asan.sh runs this file against it (the GPU objects are absent there,
    so only the host entry points are exercised)."""
    assert os.path.exists(os.environ["TMF_ASAN_LIB"])


def test_jacobi_sweep_cap_means_not_converged():
    """The Jacobi kernels report their sweep count per problem; the cap is 'rotations still pending' and must
    raise like LAPACK's non-convergence (numpy.linalg.LinAlgError), not flow on silently."""
    nat.check_jacobi_sweeps(np.array([3, 7, 59], np.int32))
    nat.check_jacobi_sweeps(np.zeros(0, np.int32))
    with pytest.raises(np.linalg.LinAlgError, match="did not converge in 60 sweeps"):
        nat.check_jacobi_sweeps(np.array([3, 60, 5], np.int32))


@pytest.mark.parametrize("name", golden_names())
def test_merged_leg_order_against_the_reference_fixtures(name):
    """The one thing `MPSData.to_tenpy` depends on that is checkable without TeNPy: the rows of the merged (p, bra) leg
    are enumerated in the order slater.py:943-952 documents (ascending left charge; inside a charge first the p = 0 rows
    of bra sector Q, then the p = 1 rows of sector Q - 1, each in the bra's own order).  For every site of every fixture
    the occupation pattern of merged row r over the sometimes-occupied orbitals - rebuilt from (bra_p[r], bra_alpha[r]),
    the bra's patterns and the native row selection - must equal row r of the REFERENCE's `new_sets_bra`; likewise the
    ket patterns against `new_sets_ket`."""
    g = load(name)
    L, oc = int(g["L"]), int(g["ortho_center"])

    def cut(b):
        sets = g[f"b{b}_sets"]
        m = np.zeros((len(sets), 2), np.uint64)
        for i in range(sets.shape[1]):
            m[:, i // 64] |= sets[:, i].astype(np.uint64) << np.uint64(i % 64)
        nfl, nfr = (int(x) for x in g[f"b{b}_nfilled"])
        return dict(sets=sets, masks=m, k=sets.shape[1], nfl=nfl, nfr=nfr, q=nfl + sets.sum(axis=1))

    for i in range(L):
        mode = 0 if i < oc else 1
        bra, ket = (cut(i), cut(i + 1)) if mode == 0 else (cut(i + 1), cut(i))
        nfb, nfk = (bra["nfl"], ket["nfl"]) if mode == 0 else (bra["nfr"], ket["nfr"])
        r = nat.site_prepare(mode, bra["k"], nfb, bra["masks"], bra["q"], ket["k"], nfk, ket["masks"], ket["q"])

        def occupied(side, orb, alpha, p):
            """occupation of orbital `orb` (numbering [entangled | filled], -1 = physical) in Schmidt vector alpha"""
            if orb < 0:
                return bool(p)
            if orb >= side["k"]:
                return True                                   # a filled orbital
            if mode == 0:
                return bool(side["sets"][alpha, orb])
            return not side["sets"][alpha, side["k"] - 1 - orb]   # slater.py:465

        ka, ref_bra, ref_ket = r["k"], g[f"s{i}_sets_bra"], g[f"s{i}_sets_ket"]
        assert ref_bra.shape == (2 * len(bra["sets"]), r["sb"]) and ref_ket.shape == (len(ket["sets"]), r["sk"])
        for row in range(len(ref_bra)):
            mine = [occupied(bra, int(o), int(r["bra_alpha"][row]), int(r["bra_p"][row])) for o in r["row_sel"][ka:]]
            assert mine == ref_bra[row].tolist(), (name, i, row)
        for col in range(len(ref_ket)):
            mine = [occupied(ket, int(o), col, 0) for o in r["col_sel"][ka:]]
            assert mine == ref_ket[col].tolist(), (name, i, col)


def test_arnoldi_dominant_eigenpair():
    """gutzwiller._dominant_eigenpair (host side of the transfer-matrix fixed points): non-normal maps, real and complex,
    dimension below and above the Krylov size, slowly separating spectrum."""
    from temfpy_amd.gutzwiller import _dominant_eigenpair

    rng = np.random.default_rng(3)
    for n, cplx, gap in ((1, False, 0.5), (3, True, 0.5), (200, False, 0.5), (300, True, 0.97)):
        V = rng.normal(size=(n, n)) + (1j * rng.normal(size=(n, n)) if cplx else 0)
        w = gap * rng.uniform(0.0, 1.0, n) * np.exp(2j * np.pi * rng.uniform(size=n) if cplx else 0)
        w[0] = 1.3
        A = (V * w) @ np.linalg.inv(V)
        if not cplx:
            A = A.real
        calls = [0]

        def apply(v):
            calls[0] += 1
            return A @ v
        th, v = _dominant_eigenpair(apply, np.ones(n, complex if cplx else float))
        assert abs(th - 1.3) < 1e-11 and np.linalg.norm(A @ v - th * v) < 1e-10 * np.linalg.norm(v)
        assert calls[0] < 60 * 24
    # a map that lives in fewer dimensions than the length of its vectors (padding entries of the packed blocks): the Krylov
    # space is exhausted before the Krylov size is reached, and the leftover rounding noise must not produce Ritz values
    n, pad = 13, 3
    V = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    w = np.concatenate(([3926.9], 1500 * rng.uniform(0, 1, n - 1) * np.exp(2j * np.pi * rng.uniform(size=n - 1))))
    A = np.zeros((n + pad, n + pad), complex)
    A[:n, :n] = (V * w) @ np.linalg.inv(V)
    v0 = np.concatenate((np.ones(n), np.zeros(pad))).astype(complex)
    calls = [0]

    def apply_padded(x):
        calls[0] += 1
        return A @ x
    th, v = _dominant_eigenpair(apply_padded, v0)
    assert abs(th - 3926.9) < 1e-8 and np.linalg.norm(A @ v - th * v) < 1e-8 * np.linalg.norm(v)
    assert calls[0] == n          # stopped when the 13-dimensional space was exhausted, not at the vector length
    # a REAL map whose dominant eigenvalues are a complex pair: no real eigenvector exists; the pair's value comes back with
    # its imaginary part (the caller demands a real positive eigenvalue and raises), not silently as its real part
    n = 40
    A = np.diag(0.3 * rng.uniform(size=n))
    A[:2, :2] = 1.1 * np.array([[np.cos(0.7), -np.sin(0.7)], [np.sin(0.7), np.cos(0.7)]])
    Q = np.linalg.qr(rng.normal(size=(n, n)))[0]
    A = Q @ A @ Q.T
    th, v = _dominant_eigenpair(lambda x: A @ x, np.ones(n))
    assert abs(abs(th) - 1.1) < 1e-9 and abs(abs(th.imag) - 1.1 * np.sin(0.7)) < 1e-9


def test_dead_states_are_removed_from_the_bonds_of_a_cell():
    """gutzwiller._drop_dead_boundary_states (host side of the infinite projection): a bond index whose row of the next tensor
    vanished disappears from that bond - rows of the tensor to its right, columns of the one to its left (periodically),
    Schmidt value and label - block ranges are renumbered and the rows that lost a column rescaled."""
    from temfpy_amd.gutzwiller import _drop_dead_boundary_states

    rng = np.random.default_rng(0)

    def iso(n, m):
        return np.linalg.qr(rng.normal(size=(m, n)))[0].T          # n orthonormal rows of length m

    # two sites, bonds of 3 and 2 states; sectors: bond 0 = labels [0, 0, 1], bond 1 = [0, 1]
    a0 = iso(2, 2)
    b0 = [(0, 0, 0, 0, 2, 0, 1, a0[:, :1].copy()), (1, 0, 1, 0, 2, 1, 2, a0[:, 1:].copy()), (0, 1, 1, 2, 3, 1, 2, np.zeros((1, 1)))]
    a1 = iso(2, 3)
    a1[:, 2] *= 1e-4                                                # the column of the dead state carries next to nothing
    b1 = [(0, 0, 0, 0, 1, 0, 2, a1[:1, :2].copy()), (1, 1, 1, 1, 2, 2, 3, a1[1:, 2:].copy()), (0, 1, 0, 1, 2, 0, 2, a1[1:, :2].copy())]
    lam = [np.array([0.8, 0.6, 1e-7]), np.array([0.9, 0.435889894]), np.array([0.8, 0.6, 1e-7])]
    ch = [np.array([0, 0, 1]), np.array([0, 1]), np.array([0, 0, 1])]
    blocks, lam2, ch2 = _drop_dead_boundary_states([b0, b1], lam, ch)
    assert [len(x) for x in lam2] == [2, 2, 2] and ch2[0].tolist() == [0, 0] and ch2[2].tolist() == [0, 0]
    assert abs(np.linalg.norm(lam2[0]) - 1) < 1e-15 and np.array_equal(lam2[0], lam2[2])
    assert all(bl[3] < 2 and bl[4] <= 2 for bl in blocks[0]) and len(blocks[0]) == 2            # the dead row's block is gone
    assert all(bl[6] <= 2 for bl in blocks[1]) and len(blocks[1]) == 2                          # and its column on the left
    w = np.zeros(2)
    for (_p, _ql, _qr, l0, l1, _r0, _r1, a) in blocks[1]:
        w[l0:l1] += (np.abs(a) ** 2).sum(axis=1)
    np.testing.assert_allclose(w, 1.0, atol=1e-14)                                              # rows rescaled
    same = _drop_dead_boundary_states([b1, b1], [lam[1], lam[1], lam[1]], [ch[1]] * 3)          # nothing to remove
    assert same[0][0] is b1 and len(same[1][0]) == 2


def test_tenpy_switch_and_back_reference():
    """as_tenpy = False / None / True on the entry points (slater._maybe_tenpy) and the reference a TeNPy object returned by the
    package carries to its device-side container (gutzwiller.native), without TeNPy installed."""
    import pytest

    from temfpy_amd.gutzwiller import native
    from temfpy_amd.slater import _maybe_tenpy

    class NoTenpy:
        L, lam = 3, [np.ones(1), np.ones(2), np.ones(2), np.ones(1)]

        def to_tenpy(self):
            raise ImportError("No module named 'tenpy'")

    class Fake:
        def __init__(self, own, L=3, chi=(2, 2)):
            self._temfpy_amd, self.L, self.chi = own, L, list(chi)

    r = NoTenpy()
    assert _maybe_tenpy(r, False) is r and _maybe_tenpy(r, None) is r
    with pytest.raises(ImportError):
        _maybe_tenpy(r, True)
    assert native(r) is r and native(Fake(r)) is r
    with pytest.raises(ValueError, match="modified"):
        native(Fake(r, L=4))
    with pytest.raises(ValueError, match="modified"):
        native(Fake(r, chi=(2, 3)))
    inf = Fake(r, chi=(1, 2, 2))           # TeNPy lists bonds 0 .. L-1 of an infinite MPS
    assert native(inf) is r


def test_infinite_cell_adapter_labels():
    """gutzwiller._fermions_from_imps (host side): the labels of the closing bond are those of bond 0 plus the charge of one
    cell (mod 2 for parity labels), so that q_l + p = q_r holds on every site; form and charge kind are checked."""
    import pytest

    from temfpy_amd.gutzwiller import _fermions_from_imps
    from temfpy_amd.iMPS import iMPSData

    a = np.arange(1.0, 5.0).reshape(2, 2)
    lam = [np.ones(2) / np.sqrt(2)] * 3
    # number labels: bond 0 = [0, 1], bond 1 = [0, 1]; one particle per cell
    blocks = [[(0, 0, 0, 0, 1, 0, 1, a[:1, :1]), (1, 0, 1, 0, 1, 1, 2, a[:1, 1:]), (0, 1, 1, 1, 2, 1, 2, a[1:, 1:])],
              [(1, 0, 0, 0, 1, 0, 1, a[:1, :1]), (0, 1, 0, 1, 2, 0, 1, a[1:, :1]), (1, 1, 1, 1, 2, 1, 2, a[1:, 1:])]]
    cell = iMPSData(blocks, lam, [np.array([0, 1])] * 3, 1, 2, conserve="N")
    f = _fermions_from_imps(cell)
    assert f.infinite and f.cell_charge == 1 and f.conserve == "N" and f.L == 2 and f.oc == 2
    assert [x.tolist() for x in f.charges] == [[0, 1], [0, 1], [1, 2]]
    for r in f.blocks:                                   # every packed block obeys q_l + p = q_r in these labels
        assert int(r["cl"]) + int(r["p"]) == int(r["cr"])
    par = iMPSData(blocks, lam, [np.array([0, 1])] * 3, 1, 2, conserve="parity")
    g = _fermions_from_imps(par)
    assert [x.tolist() for x in g.charges] == [[0, 1], [0, 1], [0, 1]] and g.perm[2].tolist() == [1, 0]      # (1, 2) mod 2, sorted
    for r in g.blocks:
        assert (int(r["cl"]) + int(r["p"])) % 2 == int(r["cr"])
    with pytest.raises(ValueError, match="FermionSite must conserve"):
        _fermions_from_imps(iMPSData(blocks, lam, [np.array([0, 1])] * 3, 1, 2, conserve="spin Sz"))
    cell.form = ["B", None]
    with pytest.raises(ValueError, match="right-canonical"):
        _fermions_from_imps(cell)


def test_site_prepare_without_physical_leg_matches_oracle():
    """tmf_site_prepare with mode bit 1 (two bases of the SAME orbitals, slater.py:1023-1024: the gauge overlaps of
    ``C_to_iMPS``, slater.py:1538): bra rows are the bra Schmidt vectors themselves (no doubling, no re-sorting), the always /
    sometimes split, signs, Schur complement and index lists against the oracle's ``site_tensor(short, long, "left")``."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_imps import ssh

    L, cell, cut, chi = 20, 2, 10, 24
    Cs, _ = orc.correlation_matrix(ssh(L))
    Cl, _ = orc.correlation_matrix(ssh(L + cell))
    tr = orc.as_trunc({"chi_max": chi})
    bra, ket = orc.cut_vectors(Cs, cut, tr, "LR"), orc.cut_vectors(Cl, cut, tr, "LR")
    for mode, side in ((0, "L"),):
        s = orc.site_tensor(bra, ket, "left")
        Vb, nfb = _device_layout(bra, side)
        Vk, nfk = _device_layout(ket, side)
        qb = bra.n_filled("L") + bra.sets.sum(axis=1)
        qk = ket.n_filled("L") + ket.sets.sum(axis=1)
        r = nat.site_prepare(mode | 2, bra.k, nfb, _masks(bra), qb, ket.k, nfk, _masks(ket), qk)
        chi_b = len(bra.lam)
        np.testing.assert_array_equal(r["bra_p"][:chi_b], 0)
        np.testing.assert_array_equal(r["bra_alpha"][:chi_b], np.arange(chi_b))
        assert (r["sb"], r["sk"]) == s.M.shape
        assert not np.any(r["row_sel"] < 0)                      # no physical orbital anywhere
        Ofull = Vb.conj().T @ Vk
        W = np.zeros((r["mb"], r["mk"]), complex)
        for a, (rs, sg) in enumerate(zip(r["row_sel"], r["row_sign"])):
            W[a] = sg * Ofull[rs, r["col_sel"]] * r["col_sign"]
        k = r["k"]
        det, M = (np.linalg.det(W[:k, :k]), W[k:, k:] - W[k:, :k] @ np.linalg.inv(W[:k, :k]) @ W[:k, k:]) if k else (1.0, W)
        np.testing.assert_allclose(det, s.det_always, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(M, s.M, rtol=0, atol=1e-10)
        assert sorted(int(x) for x in r["sectors"]["q"]) == sorted(s.blocks)
        for sec in r["sectors"]:
            r0, r1, c0, c1, blk = s.blocks[int(sec["q"])]
            assert (sec["r0"], sec["r1"], sec["c0"], sec["c1"]) == (r0, r1, c0, c1)


def test_patterns_of_equal_weight_come_in_one_order_whatever_the_rounding():
    """tmf_cut_vectors orders occupation patterns whose subset sums agree within the noise of the eigenvalues by their masks
    (DESIGN 10.5): a particle-hole symmetric spectrum (e_i, 1 - e_i: partner patterns of equal weight) perturbed by 1e-15 in
    ten different ways gives ten times the same list of patterns, also where a weak orbital (e = 1e-10: a_i uncertain by
    1e-5) makes partner weights differ by parts per million; TMF_TIE_ORDER is read once per process, so the heap order is not
    compared here."""
    rng = np.random.default_rng(12)
    half = np.array([0.31, 0.07, 2e-4, 1e-10])
    e0 = np.sort(np.concatenate((half, 1.0 - half)))[::-1]
    ref = None
    for trial in range(10):
        e = e0 + rng.uniform(-1, 1, e0.size) * 1e-15 * (trial > 0)
        e = np.clip(e, 1e-300, 1.0 - 1e-17)
        sets, lam, q, _ = nat.cut_vectors(e, 0, 60, 1e-6, 1e-12)
        if ref is None:
            ref = (sets.copy(), q.copy())
            assert len(sets) > 20
        else:
            np.testing.assert_array_equal(q, ref[1])
            np.testing.assert_array_equal(sets, ref[0])
    # inside a charge sector the weights are still descending up to that noise
    for c in np.unique(ref[1]):
        w = lam[ref[1] == c]
        assert np.all(np.diff(w) <= 1e-4 * w[:-1])


def test_group_order_of_the_centre_cut():
    """engine._group_order (= Sweep::group_order): inside runs of eigenvalues closer than the tolerance the columns go by
    descending e (1 - e) (the order of the reference's group SVD, utils.py:66-94); exact multiplets and separated eigenvalues
    stay."""
    from temfpy_amd.engine import _group_order

    e = np.array([1 - 1e-9, 1 - 3e-8, 1 - 4e-7, 0.9, 0.5, 0.2, 3e-7, 1e-8])
    assert _group_order(e, 1e-12) is None
    p = _group_order(e, 1e-6)
    np.testing.assert_array_equal(p, [2, 1, 0, 3, 4, 5, 6, 7])        # the three next to 1 reversed, the two next to 0 already in order
    assert _group_order(np.array([0.7, 0.7, 0.3, 0.3]), 1e-12) is None  # exact pairs keep their places
    p = _group_order(np.array([0.6, 0.55, 0.52]), 0.1)
    np.testing.assert_array_equal(p, [2, 1, 0])


def test_gutzwiller_table_helpers_vectorised_forms():
    """_gemm_tiles_spans / _sector_tables (all launches / bonds at once) against the per-launch / per-bond forms."""
    from temfpy_amd import gutzwiller as g, _native as nat

    rng = np.random.default_rng(0)
    spans, ds, o = [], [], 0
    for _ in range(300):
        n = int(rng.integers(0, 6))
        d = np.zeros(n, nat.gemm_desc)
        d["M"] = rng.integers(0, 300, n)
        d["N"] = rng.integers(0, 17 if rng.random() < 0.3 else 300, n)
        ds.append(d)
        spans.append((o, n))
        o += n
    t = np.concatenate(ds)
    tiles, ts, tn = g._gemm_tiles_spans(t, spans)
    for (o, n), (a, b), x in zip(spans, ts, tn):
        tl, tnn = g._gemm_tiles(t[o: o + n])
        assert (tnn == x or n == 0) and b == len(tl) and np.array_equal(tiles[a: a + b], tl)
    assert g._gemm_tiles_spans(t[:0], []) [1:] == ([], [])
    ch = [np.sort(rng.integers(-3, 4, int(rng.integers(0, 40)))) for _ in range(200)]
    assert g._sector_tables(ch) == [g._sector_table(q) for q in ch]
    assert g._sector_tables([]) == [] and g._sector_tables([np.zeros(0, int)]) == [{}]
    L = g._Launches(nat.qr_desc)
    a, b = np.zeros(2, nat.qr_desc), np.zeros(3, nat.qr_desc)
    a["m"], b["n"] = [1, 2], [3, 4, 5]
    L.add(a), L.add(b)
    tab = L.table()
    assert tab.dtype == nat.qr_desc and tab["m"].tolist() == [1, 2, 0, 0, 0] and tab["n"].tolist() == [0, 0, 3, 4, 5]


def test_assert_nambu_leaves_its_input_alone_and_blocked_comparison():
    """pfaffian.assert_nambu (pfaffian.py:189-286) forms HT(C) once as a row-major COPY (for a column-major input C.T is row-major
    already, and a conjugation in place would have hit the caller's matrix), Hermitises, and fixes the real parts in place on
    its own array; testing._all_close = np.all(|a - d| <= atol + rtol |d|) in blocks of rows, also where only the modulus of
    a complex difference decides."""
    from temfpy_amd import pfaffian, testing as T

    rng = np.random.default_rng(3)
    n = 40
    A = rng.standard_normal((2 * n, 2 * n))
    H = 1j * (A - A.T)                                   # Majorana-basis Hamiltonian
    for order in ("C", "F"):
        X = np.array(H, order=order)
        keep = X.copy()
        out = pfaffian.assert_nambu(X, "M", offset=0, name="Hamiltonian")
        assert np.array_equal(X, keep) and np.array_equal(out, (H + H.conj().T) / 2)
        Xr = np.array(A + A.T, order=order)
        keep = Xr.copy()
        out = pfaffian.assert_nambu(Xr, None, offset=0, name="matrix")
        assert np.array_equal(Xr, keep) and out.dtype == np.float64 and np.array_equal(out, A + A.T)
    a = rng.standard_normal((1024, 300)) + 1j * rng.standard_normal((1024, 300))
    for d, rt, at in ((a + 1e-12, 0, 1e-10), (a + 1e-12, 0, 1e-13), (a + 1e-12, 1e-9, 0), (a + 1e-12, 0, 1.2e-12)):
        assert T._all_close(a, d, rt, at) == bool(np.all(np.abs(a - d) <= at + rt * np.abs(d)))
    d = a.copy()
    d[700, 3] += 1e-10 * (0.8 + 0.8j)                    # |difference| = 1.13e-10, real and imaginary part 0.8e-10 each
    assert not T._all_close(a, d, 0, 1e-10)
    d = a.copy()
    d[5, 5] = np.nan
    assert not T._all_close(a, d, 0, 1e-10)
