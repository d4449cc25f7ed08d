"""GPU parity tests of every device entry point of the C ABI against NumPy (fp64).
Tolerances are stated per test; everything is called through libtemfpy_hip.so."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def eng():
    from temfpy_amd import _native as nat
    from temfpy_amd.engine import Engine

    e = Engine("cuda:0")
    e.nat = nat
    return e


def setup(eng, cplx):
    eng.dtype = eng.nat.TMF_C128 if cplx else eng.nat.TMF_F64
    eng.elem = 16 if cplx else 8
    eng._keep.clear()


def rnd(rng, shape, cplx):
    a = rng.standard_normal(shape)
    return a + 1j * rng.standard_normal(shape) if cplx else a


def dev(eng, a):
    """column-major upload; returns (tensor, ptr)"""
    t = torch.from_numpy(np.asfortranarray(a).T.copy().reshape(-1)).to("cuda:0")
    return t, t.data_ptr()


def back(t, shape):
    return t.cpu().numpy().reshape(shape[::-1]).T


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("opA", [0, 1])
def test_gemm_batched(eng, cplx, opA):
    setup(eng, cplx)
    rng = np.random.default_rng(1)
    shapes = [(1, 1, 1), (64, 64, 16), (65, 17, 33), (130, 64, 257), (200, 16, 70), (33, 3, 5), (512, 70, 300), (7, 100, 0)]
    for only16 in (False, True):
        As, Bs, Cs, keep = [], [], [], []
        sh = [(M, min(N, 16) if only16 else N, K) for M, N, K in shapes]
        for M, N, K in sh:
            A = rnd(rng, (K, M) if opA else (M, K), cplx)
            B = rnd(rng, (K, N), cplx)
            C0 = rnd(rng, (M, N), cplx)
            As.append(A), Bs.append(B), Cs.append(C0)
        for alpha, beta in ((1.0, 0.0), (-1.0, 1.0)):
            dA = [dev(eng, a) for a in As]
            dB = [dev(eng, b) for b in Bs]
            dC = [dev(eng, c) for c in Cs]
            eng.gemm(opA, alpha, beta, [x[1] for x in dA], [x[1] for x in dB], [x[1] for x in dC],
                     [s[0] for s in sh], [s[1] for s in sh], [s[2] for s in sh],
                     [max(a.shape[0], 1) for a in As], [max(b.shape[0], 1) for b in Bs], [s[0] for s in sh])
            torch.cuda.synchronize()
            for (M, N, K), A, B, C0, dc in zip(sh, As, Bs, Cs, dC):
                ref = alpha * ((A.conj().T if opA else A) @ B) + beta * C0
                got = back(dc[0], (M, N))
                np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(1, K))


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("cholqr", [False, True])
def test_bcgs2_orthonormalises_and_keeps_span(eng, cplx, cholqr):
    """tmf_bcgs_batched with both panel methods (LDS Gram-Schmidt panel kernel / Cholesky-QR twice)."""
    setup(eng, cplx)
    rng = np.random.default_rng(2)
    cases = [(40, 0, 40), (300, 0, 64), (513, 5, 70), (100, 10, 11), (1000, 0, 33), (17, 0, 1)]
    mats, dm = [], []
    for n, c0, c1 in cases:
        A = rnd(rng, (n, c1), cplx)
        if c0:
            A[:, :c0] = np.linalg.qr(A[:, :c0])[0]
        mats.append(A)
        dm.append(dev(eng, A))
    scr = eng._alloc(len(cases) * 80 * 16)
    eng.bcgs2([d[1] for d in dm], [c[0] for c in cases], [c[0] for c in cases], [c[1] for c in cases],
              [c[2] for c in cases], scr.data_ptr() + np.arange(len(cases)) * 80 * 16 * eng.elem, cholqr=cholqr)
    torch.cuda.synchronize()
    for (n, c0, c1), A, d in zip(cases, mats, dm):
        Q = back(d[0], (n, c1))
        np.testing.assert_allclose(Q.conj().T @ Q, np.eye(c1), rtol=0, atol=5e-14)
        # same column space: projector onto span(A) equals Q Q^H
        Qa = np.linalg.qr(A)[0]
        np.testing.assert_allclose(Q @ Q.conj().T, Qa @ Qa.conj().T, rtol=0, atol=1e-11)
        if c0:
            np.testing.assert_allclose(Q[:, :c0], A[:, :c0], rtol=0, atol=0)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("cholqr", [False, True])
def test_bcgs2_exactly_rank_deficient_gives_zero_columns(eng, cplx, cholqr):
    setup(eng, cplx)
    rng = np.random.default_rng(3)
    A = np.zeros((50, 8), complex if cplx else float)
    A[:, :3] = rnd(rng, (50, 3), cplx)
    d = dev(eng, A)
    scr = eng._alloc(8 * 16)
    eng.bcgs2([d[1]], [50], [50], [0], [8], np.array([scr.data_ptr()]), cholqr=cholqr)
    torch.cuda.synchronize()
    Q = back(d[0], (50, 8))
    G = Q.conj().T @ Q
    np.testing.assert_allclose(G[:3, :3], np.eye(3), atol=1e-14)
    assert np.all(Q[:, 3:] == 0)


@pytest.mark.parametrize("cplx", [True, False])
def test_jacobi_svd_and_threshold(eng, cplx):
    setup(eng, cplx)
    rng = np.random.default_rng(4)
    ps = [1, 2, 3, 17, 32, 64]
    Xs = []
    for p in ps:
        X = rnd(rng, (p, p), cplx)
        if p >= 17:  # column-graded: singular values down to 1e-9
            X = np.linalg.qr(X)[0] * np.logspace(0, -9, p) @ np.linalg.qr(rnd(rng, (p, p), cplx))[0]
        Xs.append(X)
    dX = [dev(eng, x) for x in Xs]
    dV = [dev(eng, np.zeros_like(x)) for x in Xs]
    ds = [torch.zeros(p, dtype=torch.float64, device="cuda:0") for p in ps]
    dc = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    thr2 = 2e-12  # between two singular values of the graded spectra
    eng.jacobi([d[1] for d in dX], [d[1] for d in dV], [s.data_ptr() for s in ds],
               dc.data_ptr() + 4 * np.arange(len(ps)), thr2, ps, ps, ps)
    torch.cuda.synchronize()
    cnt = dc.cpu().numpy()
    for p, X, dv, s_, c in zip(ps, Xs, dV, ds, cnt):
        V = back(dv[0], (p, p))
        s = s_.cpu().numpy()
        sref = np.linalg.svd(X, compute_uv=False)
        np.testing.assert_allclose(s, sref, rtol=1e-10, atol=1e-15)  # relative accuracy on graded spectra
        assert c == np.sum(sref**2 >= thr2)
        Vk = V[:, :c]
        np.testing.assert_allclose(Vk.conj().T @ Vk, np.eye(c), atol=1e-13)
        assert np.all(V[:, c:] == 0)
        XV = X @ Vk
        np.testing.assert_allclose(np.linalg.norm(XV, axis=0), s[:c], rtol=1e-10)
        G = XV.conj().T @ XV
        off = G - np.diag(np.diag(G))
        assert np.abs(off).max() <= 1e-13 * s[0] ** 2


@pytest.mark.parametrize("cplx", [True, False])
def test_svd_left_on_graded_triangular_factor(eng, cplx):
    """tmf_svd_left_batched on R^H (R = triangular factor of a matrix with singular values down to
    1e-15, as the range finder produces): singular values to relative accuracy, orthonormal left
    vectors spanning the same directions as numpy's SVD, columns below the threshold zeroed."""
    setup(eng, cplx)
    rng = np.random.default_rng(14)
    ps = [1, 5, 33, 64]
    Xs = []
    for p in ps:
        sv = np.logspace(-0.3, -15, p)
        B = (np.linalg.qr(rnd(rng, (3 * p, p), cplx))[0] * sv) @ np.linalg.qr(rnd(rng, (p, p), cplx))[0]
        Xs.append(np.ascontiguousarray(np.linalg.qr(B)[1].conj().T))
    dX = [dev(eng, x) for x in Xs]
    dU = [dev(eng, np.zeros_like(x)) for x in Xs]
    ds = [torch.zeros(p, dtype=torch.float64, device="cuda:0") for p in ps]
    dc = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    thr2 = 2.5e-13
    eng.jacobi([d[1] for d in dX], [d[1] for d in dU], [s.data_ptr() for s in ds],
               dc.data_ptr() + 4 * np.arange(len(ps)), thr2, ps, ps, ps, left_only=True)
    torch.cuda.synchronize()
    cnt = dc.cpu().numpy()
    for p, X, du, s_, c in zip(ps, Xs, dU, ds, cnt):
        U = back(du[0], (p, p))
        s = s_.cpu().numpy()
        Ur, sref, _ = np.linalg.svd(X)
        c_ref = int(np.sum(sref**2 >= thr2))
        assert c == c_ref
        np.testing.assert_allclose(s[:c], sref[:c], rtol=1e-9)
        Uk = U[:, :c]
        np.testing.assert_allclose(Uk.conj().T @ Uk, np.eye(c), atol=1e-12)
        assert np.all(U[:, c:] == 0)
        # same directions up to a phase (the kept singular values are well separated)
        ov = np.abs(np.einsum("ij,ij->j", Ur[:, :c].conj(), Uk))
        np.testing.assert_allclose(ov, 1.0, atol=1e-9)


@pytest.mark.parametrize("cplx", [True, False])
def test_jacobi_hermitian_eigenproblem(eng, cplx):
    setup(eng, cplx)
    rng = np.random.default_rng(5)
    p = 40
    U = np.linalg.qr(rnd(rng, (p, p), cplx))[0]
    e = np.sort(np.concatenate((1 - np.logspace(-12, -1, 15), np.logspace(-12, -1, 15), rng.uniform(0.2, 0.8, 10))))[::-1]
    T = (U * e) @ U.conj().T
    dX, dV = dev(eng, T), dev(eng, np.zeros_like(T))
    ds = torch.zeros(p, dtype=torch.float64, device="cuda:0")
    eng.jacobi([dX[1]], [dV[1]], [ds.data_ptr()], 0, 0.0, [p], [p], [p])
    torch.cuda.synchronize()
    V, s = back(dV[0], (p, p)), ds.cpu().numpy()
    # T itself carries ~p*eps construction error; compare with LAPACK on the same matrix
    np.testing.assert_allclose(s, np.linalg.eigvalsh(T)[::-1], rtol=0, atol=3e-14)
    np.testing.assert_allclose(V.conj().T @ V, np.eye(p), atol=1e-13)
    np.testing.assert_allclose(T @ V, V * s, atol=3e-14)


@pytest.mark.parametrize("cplx", [True, False])
def test_lu_schur(eng, cplx):
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(6)
    cases = [(5, 4, 0), (1, 3, 1), (20, 23, 7), (70, 66, 40), (300, 290, 257), (33, 33, 33), (48, 50, 16)]
    Ws, dW, dS = [], [], []
    for mb, mk, k in cases:
        W = rnd(rng, (mb, mk), cplx)
        Ws.append(W), dW.append(dev(eng, W))
        dS.append(dev(eng, np.zeros((max(mb - k, 1), max(mk - k, 1)), W.dtype)))
    ddet = eng._alloc(len(cases))
    sd = np.zeros(len(cases), nat.schur_desc)
    sd["W"] = [d[1] for d in dW]
    sd["S"] = [d[1] for d in dS]
    sd["det"] = ddet.data_ptr() + np.arange(len(cases)) * eng.elem
    sd["mb"], sd["mk"], sd["k"] = [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases]
    sd["ldw"] = sd["mb"]
    sd["lds"] = np.maximum(sd["mb"] - sd["k"], 1)
    t = eng._up(sd)
    nat.check(eng.lib.tmf_lu_schur_batched(eng.dtype, t.data_ptr(), len(cases), 300, eng.stream), "lu")
    torch.cuda.synchronize()
    det = ddet.cpu().numpy()
    for i, ((mb, mk, k), W) in enumerate(zip(cases, Ws)):
        if k:
            dref = np.linalg.det(W[:k, :k])
            Sref = W[k:, k:] - W[k:, :k] @ np.linalg.solve(W[:k, :k], W[:k, k:])
        else:
            dref, Sref = 1.0, W
        np.testing.assert_allclose(det[i], dref, rtol=1e-10)
        if mb > k and mk > k:
            S = back(dS[i][0], (mb - k, mk - k))
            scale = max(1.0, np.abs(Sref).max())
            np.testing.assert_allclose(S, Sref, rtol=0, atol=1e-10 * scale)
            # in-place Schur complement is what the sweep reads
            Wd = back(dW[i][0], (mb, mk))
            np.testing.assert_allclose(Wd[k:, k:], Sref, rtol=0, atol=1e-10 * scale)


@pytest.mark.parametrize("cplx", [True, False])
def test_lu_blocked_over_launches(eng, cplx):
    """tmf_lu_block_batched + tmf_lu_trsm_batched + the MFMA GEMM (the multi-launch form of tmf_lu_schur_batched the
    sweep uses): det(A) and the Schur complement D - C A^-1 B against NumPy (slater.py:1077-1090), for always-blocks
    from 0 to 4 outer steps, ragged last blocks, k = mb and k = mk."""
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(16)
    cases = [(5, 4, 0), (1, 3, 1), (20, 23, 7), (70, 66, 40), (300, 290, 257), (33, 33, 33), (48, 50, 16), (130, 140, 64),
             (131, 139, 65), (200, 210, 128), (90, 64, 64)]
    cases.sort(key=lambda c: -c[2])          # the driver's order: largest always-block first
    Ws = [rnd(rng, (mb, mk), cplx) for mb, mk, k in cases]
    dW = [dev(eng, W) for W in Ws]
    ddet = eng._alloc(len(cases))
    d_piv = torch.zeros(sum(c[2] for c in cases) + 1, dtype=torch.int32, device="cuda")
    d_T = eng._alloc(sum(64 * c[1] for c in cases) + 1)
    ld = np.zeros(len(cases), nat.lublock_desc)
    ld["W"] = [d[1] for d in dW]
    ld["det"] = ddet.data_ptr() + np.arange(len(cases)) * eng.elem
    ks = np.array([c[2] for c in cases])
    ld["piv"] = d_piv.data_ptr() + 4 * (np.cumsum(ks) - ks)
    tk = np.array([64 * c[1] for c in cases])
    ld["T"] = d_T.data_ptr() + eng.elem * (np.cumsum(tk) - tk)
    ld["mb"], ld["mk"], ld["k"] = [c[0] for c in cases], [c[1] for c in cases], ks
    ld["ldw"] = ld["mb"]
    t = eng._up(ld)
    mb, mk = ld["mb"].astype(np.int64), ld["mk"].astype(np.int64)
    for j0 in range(0, max(int(ks.max()), 1), 64):
        nact = int((ks > j0).sum())
        nat.check(eng.lib.tmf_lu_block_batched(eng.dtype, t.data_ptr(), len(cases) if j0 == 0 else nact, j0, 64,
                                               int(mb.max() if j0 == 0 else mb[:nact].max()), eng.stream), "lu_block")
        if nact == 0:
            break
        cend = np.minimum(ks[:nact], j0 + 64)
        nat.check(eng.lib.tmf_lu_trsm_batched(eng.dtype, t.data_ptr(), nact, j0, 64, int((mk[:nact] - cend).max()), eng.stream),
                  "lu_trsm")
        el = eng.elem
        eng.gemm(0, -1.0, 1.0, ld["W"][:nact] + ((cend + j0 * mb[:nact]) * el).astype(np.uint64), ld["T"][:nact],
                 ld["W"][:nact] + ((cend + cend * mb[:nact]) * el).astype(np.uint64), mb[:nact] - cend, mk[:nact] - cend,
                 cend - j0, mb[:nact], np.full(nact, 64), mb[:nact])
    torch.cuda.synchronize()
    det = ddet.cpu().numpy()
    for i, ((mb_, mk_, k), W) in enumerate(zip(cases, Ws)):
        if k:
            dref = np.linalg.det(W[:k, :k])
            Sref = W[k:, k:] - W[k:, :k] @ np.linalg.solve(W[:k, :k], W[:k, k:])
        else:
            dref, Sref = 1.0, W
        np.testing.assert_allclose(det[i], dref, rtol=1e-10)
        if mb_ > k and mk_ > k:
            Wd = back(dW[i][0], (mb_, mk_))
            scale = max(1.0, np.abs(Sref).max())
            np.testing.assert_allclose(Wd[k:, k:], Sref, rtol=0, atol=1e-10 * scale)


@pytest.mark.parametrize("cplx", [True, False])
def test_block_local_elimination(eng, cplx):
    """tmf_diag_inverse_batched + two MFMA GEMM launches per outer step (the sweep's default Schur-complement path): det(A),
    the Schur complement D - C A^-1 B (slater.py:1077-1090) and the reported statistics, on matrices whose always-block
    is block-diagonally dominant with arbitrary (pivoting needed) diagonal blocks; a matrix whose diagonal block is
    singular must report a zero pivot."""
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(21)
    cases = [(5, 4, 0), (1, 3, 1), (20, 23, 7), (70, 66, 40), (300, 290, 257), (33, 33, 33), (48, 50, 16), (130, 140, 64),
             (131, 139, 65), (200, 210, 128), (90, 64, 64), (150, 160, 100)]
    cases.sort(key=lambda c: -c[2])
    Ws = []
    for mb, mk, k in cases:
        W = 0.05 * rnd(rng, (mb, mk), cplx)
        b0 = 0
        while b0 < k:                                                  # the kernel's blocks: (k - 1) % 64 + 1 columns, then 64 each
            b1 = b0 + ((k - 1) % 64 + 1 if b0 == 0 else 64)
            W[b0:b1, b0:b1] = rnd(rng, (b1 - b0, b1 - b0), cplx)       # no structure inside a block: pivoting is needed there
            b0 = b1
        Ws.append(W)
    singular = cases.index((20, 23, 7))     # second round: its only diagonal block made rank deficient
    for bad in (False, True):
        if bad:
            Ws[singular][:7, :7] = 0.0
            Ws[singular][:7, :7][0, 0] = 1.0
        dW = [dev(eng, W) for W in Ws]
        n = len(cases)
        ddet = eng._alloc(n)
        d_inv = eng._alloc(64 * 64 * n)
        tk = np.array([64 * c[1] for c in cases])
        d_X = eng._alloc(int(tk.sum()) + 1)
        d_stats = torch.zeros(2 * n, dtype=torch.float64, device="cuda")   # per matrix: min |pivot|^2, max |D^-1 entry|^2
        ld = np.zeros(n, nat.diaginv_desc)
        el = eng.elem
        ld["W"] = [d[1] for d in dW]
        ld["det"] = ddet.data_ptr() + np.arange(n) * el
        ld["inv"] = d_inv.data_ptr() + el * 64 * 64 * np.arange(n)
        X = d_X.data_ptr() + el * (np.cumsum(tk) - tk)
        ks = np.array([c[2] for c in cases])
        ld["mb"], ld["mk"], ld["k"] = [c[0] for c in cases], [c[1] for c in cases], ks
        ld["ldw"] = ld["mb"]
        t = eng._up(ld)
        mb, mk = ld["mb"].astype(np.int64), ld["mk"].astype(np.int64)
        for step in range(max(-(-int(ks.max()) // 64), 1)):
            nact = int((ks > 64 * step).sum())
            nat.check(eng.lib.tmf_diag_inverse_batched(eng.dtype, t.data_ptr(), n if step == 0 else nact, step, d_stats.data_ptr(),
                                                       eng.stream), "diag_inverse")
            if nact == 0:
                break
            nb0 = (ks[:nact] - 1) % 64 + 1                  # blocks counted from the end: a ragged first one
            j0 = np.zeros(nact, np.int64) if step == 0 else nb0 + 64 * (step - 1)
            cend = nb0 if step == 0 else j0 + 64
            nb, rows2, cols2 = cend - j0, mb[:nact] - cend, mk[:nact] - cend
            use = (rows2 > 0) & (cols2 > 0)
            if not use.any():
                continue
            u64 = np.uint64
            A12 = ld["W"][:nact] + ((j0 + cend * mb[:nact]) * el).astype(u64)
            A21 = ld["W"][:nact] + ((cend + j0 * mb[:nact]) * el).astype(u64)
            W22 = ld["W"][:nact] + ((cend + cend * mb[:nact]) * el).astype(u64)
            w64 = np.full(int(use.sum()), 64)
            eng.gemm(0, 1.0, 0.0, ld["inv"][:nact][use], A12[use], X[:nact][use], nb[use], cols2[use], nb[use], w64, mb[:nact][use], w64)
            eng.gemm(0, -1.0, 1.0, A21[use], X[:nact][use], W22[use], rows2[use], cols2[use], nb[use], mb[:nact][use], w64, mb[:nact][use])
        torch.cuda.synchronize()
        det = ddet.cpu().numpy()
        stats = np.sqrt(d_stats.cpu().numpy().reshape(n, 2))
        if bad:
            assert stats[singular, 0] == 0.0
            continue
        assert 1e-3 < stats[ks > 0, 0].min() and stats[ks > 0, 0].max() < 3.0, stats
        # the inverse statistic only covers blocks with always-rows below them
        assert np.all(stats[ks <= 64, 1] == 0.0) and 0.0 < stats[ks > 64, 1].min() and stats[:, 1].max() < 100.0, stats
        for i, ((mb_, mk_, k), W) in enumerate(zip(cases, Ws)):
            if k:
                dref = np.linalg.det(W[:k, :k])
                Sref = W[k:, k:] - W[k:, :k] @ np.linalg.solve(W[:k, :k], W[:k, k:])
            else:
                dref, Sref = 1.0, W
            np.testing.assert_allclose(det[i], dref, rtol=1e-9)
            if mb_ > k and mk_ > k:
                Wd = back(dW[i][0], (mb_, mk_))
                scale = max(1.0, np.abs(Sref).max())
                np.testing.assert_allclose(Wd[k:, k:], Sref, rtol=0, atol=1e-9 * scale)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("n,cls", [(0, 0), (1, 1), (2, 2), (5, 5), (8, 8), (9, 9), (12, 12), (13, 13), (16, 16), (17, 17), (19, 19), (25, 25), (32, 32), (40, 64),
                                   (7, 255), (40, 255), (70, 255), (90, 255)])
def test_det_gather(eng, cplx, n, cls):
    """tmf_det_gather_batched against numpy.linalg.det (slater.py:828-869).  Class 255: the sometimes-matrix stays in global
    memory (here 120 x 110: 211 KB complex, beyond any LDS stage) and only the minor, of order up to 90, is held in LDS."""
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(7 + n)
    sb, sk = max(n + 6, 3), max(n + 4, 2)
    if cls == 255:
        sb, sk = 120, 110
    elif sb * sk * eng.elem > 60000:
        sb = sk = n + 2
    S = rnd(rng, (sb, sk), cplx)
    nsb, nsk = 37, 21 if n < 33 else 5
    if cls == 255:
        nsb, nsk = 9, 4
    bra = np.stack([np.sort(rng.choice(sb, n, replace=False)) for _ in range(nsb)]).astype(np.uint8).reshape(nsb, n)
    ket = np.stack([np.sort(rng.choice(sk, n, replace=False)) for _ in range(nsk)]).astype(np.uint8).reshape(nsk, n)
    scale = rnd(rng, (1,), cplx)
    dS_, dsc = dev(eng, S), dev(eng, scale)
    tb, tk = eng._up(bra if n else np.zeros(1, np.uint8)), eng._up(ket if n else np.zeros(1, np.uint8))
    out = eng._alloc(nsb * nsk, zero=True)
    ta = 16
    dd = np.zeros(_cdiv(nsb, ta), nat.det_desc)
    for j in range(len(dd)):
        dd[j] = (dS_[1], dsc[1], tb.data_ptr(), tk.data_ptr(), out.data_ptr(), sb, sk, sb, n, nsb, nsk, j * ta,
                 min(nsb, (j + 1) * ta))
    a16 = lambda x: (x + 15) & ~15  # noqa: E731
    gpw = 8 if n <= 8 else 4 if n <= 16 else 2
    lds = a16(sb * sk * eng.elem) + a16(nsk * n) + a16(ta * n) + (n * n * eng.elem if cls == 64 else 4 * ((n | 1) * sk + gpw * (n + 1)) * eng.elem) + 16
    if cls == 255:
        lds = a16(n * n * eng.elem) + 64
    t = eng._up(dd)
    nat.check(eng.lib.tmf_det_gather_batched(eng.dtype, cls, t.data_ptr(), len(dd), lds, eng.stream), "det")
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(nsb, nsk)
    ref = np.empty((nsb, nsk), S.dtype)
    for a in range(nsb):
        for b in range(nsk):
            ref[a, b] = scale[0] * (np.linalg.det(S[np.ix_(bra[a], ket[b])]) if n else 1.0)
    np.testing.assert_allclose(got, ref, rtol=1e-9 if n < 64 else 1e-8, atol=1e-12 * np.abs(ref).max())


def _cdiv(a, b):
    return (a + b - 1) // b


def test_det_gather_singular_minor_is_zero(eng):
    setup(eng, True)
    nat = eng.nat
    S = np.ones((4, 4), complex)  # every 2x2 minor is singular
    bra = np.array([[0, 1], [2, 3]], np.uint8)
    ket = np.array([[0, 2], [1, 3]], np.uint8)
    dS_, dsc = dev(eng, S), dev(eng, np.array([2.0 + 0j]))
    tb, tk = eng._up(bra), eng._up(ket)
    out = eng._alloc(4)
    dd = np.zeros(1, nat.det_desc)
    dd[0] = (dS_[1], dsc[1], tb.data_ptr(), tk.data_ptr(), out.data_ptr(), 4, 4, 4, 2, 2, 2, 0, 2)
    t = eng._up(dd)
    nat.check(eng.lib.tmf_det_gather_batched(eng.dtype, 2, t.data_ptr(), 1, 8192, eng.stream), "det")
    torch.cuda.synchronize()
    assert np.all(out.cpu().numpy() == 0)


@pytest.mark.parametrize("cplx", [True, False])
def test_utilities(eng, cplx):
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(8)
    n = 77
    A = rnd(rng, (n, n), cplx)
    t_in = torch.from_numpy(A.reshape(-1).copy()).to("cuda:0")
    t_out = eng._alloc(n * n)
    nat.check(eng.lib.tmf_transpose(eng.dtype, t_in.data_ptr(), t_out.data_ptr(), n, eng.stream), "tr")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(back(t_out, (n, n)), A)
    # normal fill: deterministic, unit variance
    r1, r2 = eng._alloc(100001), eng._alloc(100001)
    for r in (r1, r2):
        nat.check(eng.lib.tmf_fill_normal(eng.dtype, r.data_ptr(), 100001, 42, eng.stream), "fill")
    torch.cuda.synchronize()
    x = r1.cpu().numpy()
    assert np.array_equal(x, r2.cpu().numpy())
    xr = x.view(np.float64)
    assert abs(xr.mean()) < 0.01 and abs(xr.std() - 1) < 0.01
    # column normalise with reversal and odd flip (slater.py:410)
    B = rnd(rng, (50, 6), cplx)
    dB, dD = dev(eng, B), dev(eng, np.zeros_like(B))
    eng.colcopy([dB[1]], [dD[1]], [50], [6], [50], [50], reverse=1, flip_odd=1)
    torch.cuda.synchronize()
    ref = (B / np.linalg.norm(B, axis=0))[:, ::-1].copy()
    ref[:, 1::2] *= -1
    np.testing.assert_allclose(back(dD[0], (50, 6)), ref, atol=1e-15)


@pytest.mark.parametrize("cplx", [True, False])
def test_bcgs2_numerically_rank_deficient_square_slab(eng, cplx):
    """p = n columns spanning a numerically rank-deficient space (singular values down to 1e-17):
    re-orthogonalised rounding noise must not be normalised into non-orthogonal unit vectors."""
    setup(eng, cplx)
    rng = np.random.default_rng(11)
    n = 38
    U, V = np.linalg.qr(rnd(rng, (n, n), cplx))[0], np.linalg.qr(rnd(rng, (58, 58), cplx))[0]
    sv = np.concatenate((np.logspace(-0.3, -16.5, 30), np.zeros(n - 30)))
    F = (U * sv) @ V[:n]
    Y = F @ rnd(rng, (58, n), cplx)
    d = dev(eng, Y)
    scr = eng._alloc(n * 16)
    eng.bcgs2([d[1]], [n], [n], [0], [n], np.array([scr.data_ptr()]), passes=3)
    torch.cuda.synchronize()
    Q = back(d[0], (n, n))
    nz = np.linalg.norm(Q, axis=0) > 0
    assert 25 <= nz.sum() < n                       # rounding-noise columns became exact zeros
    Qk = Q[:, nz]
    G = Qk.conj().T @ Qk
    # strong directions orthonormal to working precision; columns kept just above the 1e-14 drop
    # threshold (directions 7 decades below the physical 1e-6 cut) to 1e-8
    np.testing.assert_allclose(G[:20, :20], np.eye(20), atol=1e-12)
    np.testing.assert_allclose(G, np.eye(nz.sum()), atol=1e-8)
    # every direction with singular value >= 1e-9 is captured to ~1e-16 / 1e-9
    big = U[:, sv >= 1e-9]
    np.testing.assert_allclose(Qk @ (Qk.conj().T @ big), big, atol=1e-6)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("n,sk,mode", [(1, 5, "near"), (2, 9, "near"), (5, 12, "near"), (8, 20, "near"), (12, 30, "near"),
                                       (13, 31, "far"), (16, 40, "near"), (19, 45, "near"), (20, 64, "far"),
                                       (32, 64, "near"), (12, 30, "rankdef"), (6, 14, "graded"),
                                       (12, 40, "mixed"), (7, 30, "mixed"), (24, 60, "mixed")])
def test_det_reduced_matches_numpy(eng, cplx, n, sk, mode):
    """tmf_det_reduced_batched (one pivoted Gauss-Jordan per bra row-set, order-d minors) against
    numpy.linalg.det of every minor.  'near': ket sets differ from the first one in <= 3 columns
    (the situation of a sweep); 'far': random ket sets (up to n exchanged columns, the > 8 path);
    'rankdef': some bra row-sets select a rank-deficient slab (all their minors are 0);
    'graded': rows scaled over 12 decades (weak orbitals); 'mixed': 150 ket sets, up to 6 exchanges
    in sequence (fast lane-per-pair path and the queued > 3 path interleaved over several sweeps)."""
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(100 + n)
    sb = n + 7
    S = rnd(rng, (sb, sk), cplx)
    if mode == "graded":
        S *= np.logspace(0, -12, sb)[:, None]
    if mode == "rankdef":
        S[3] = 2.0 * S[1] - S[2]          # rows 1, 2, 3 linearly dependent
    nsb, nsk = (23, 41) if mode != "mixed" else (9, 150)
    bra = np.stack([np.sort(rng.choice(sb, n, replace=False)) for _ in range(nsb)]).astype(np.uint8)
    if mode == "rankdef":
        for a_ in (0, 5):  # these row-sets contain the dependent rows 1, 2, 3
            rest = np.setdiff1d(rng.choice(sb, sb, replace=False), [1, 2, 3])[: n - 3]
            bra[a_] = np.sort(np.concatenate(([1, 2, 3], rest)))
    base = np.sort(rng.choice(sk, n, replace=False))
    ket = []
    for _ in range(nsk):
        if mode == "far":
            ket.append(np.sort(rng.choice(sk, n, replace=False)))
        else:
            k = base.copy()
            for _ in range(rng.integers(0, min(6 if mode == "mixed" else 3, n, sk - n) + 1)):
                free = np.setdiff1d(np.arange(sk), k)
                k[rng.integers(n)] = rng.choice(free)
            ket.append(np.sort(k))
    ket[0] = base
    ket = np.stack(ket).astype(np.uint8)
    scale = rnd(rng, (1,), cplx)
    dS_, dsc = dev(eng, S), dev(eng, scale)
    tb, tk = eng._up(bra), eng._up(ket)
    out = eng._alloc(nsb * nsk, zero=True)
    ta = 8
    dd = np.zeros(_cdiv(nsb, ta), nat.det_desc)
    for j in range(len(dd)):
        dd[j] = (dS_[1], dsc[1], tb.data_ptr(), tk.data_ptr(), out.data_ptr(), sb, sk, sb, n, nsb, nsk, j * ta,
                 min(nsb, (j + 1) * ta))
    a16 = lambda x: (x + 15) & ~15  # noqa: E731
    lds = nat.reduced_det_lds(eng.elem, n, sb, sk, nsk, ta)
    if lds > 160 * 1024:
        pytest.skip("tile exceeds the 160 KiB LDS; the engine uses tmf_det_gather_batched for it")
    t = eng._up(dd)
    nat.check(eng.lib.tmf_det_reduced_batched(eng.dtype, n, t.data_ptr(), len(dd), lds, eng.stream), "red")
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(nsb, nsk)
    ref = np.empty((nsb, nsk), S.dtype)
    for a in range(nsb):
        for b in range(nsk):
            ref[a, b] = scale[0] * np.linalg.det(S[np.ix_(bra[a], ket[b])])
    if mode == "graded":  # every row of `ref` has its own scale
        for a in range(nsb):
            np.testing.assert_allclose(got[a], ref[a], rtol=0, atol=1e-9 * np.abs(ref[a]).max())
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    if mode == "rankdef":
        assert np.abs(got[0]).max() <= 1e-10 * np.abs(ref).max() and np.abs(got[5]).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("n1,n2", [(0, 0), (1, 1), (2, 0), (0, 2), (3, 1), (4, 4), (5, 3), (7, 7), (8, 8), (9, 11), (16, 16)])
def test_pf_gather(eng, cplx, n1, n2):
    """tmf_pf_gather_batched against the oracle's Parlett-Reid Pfaffian on every gathered skew
    sub-matrix (index order: ket positions then bra positions, pfaffian.py:1468-1473)."""
    from oracle import pfaffian_oracle as porc

    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(50 + 7 * n1 + n2)
    nb_, nk_ = n1 + 5, n2 + 6           # bra / ket mode counts; N = [[BB, BA], [-BA^T, AA]] (pfaffian.py:1400)
    nn = nb_ + nk_
    M = rnd(rng, (nn, nn), cplx)
    N = M - M.T
    nsb, nsk = 19, 13
    # ket positions live in [0, nk_), bra positions in [nk_, nn)
    ket = np.stack([np.sort(rng.choice(nk_, n2, replace=False)) for _ in range(nsk)]).astype(np.uint8).reshape(nsk, n2)
    bra = np.stack([nk_ + np.sort(rng.choice(nb_, n1, replace=False)) for _ in range(nsb)]).astype(np.uint8).reshape(nsb, n1)
    scale = rnd(rng, (1,), cplx)
    dN, dsc = dev(eng, N), dev(eng, scale)
    tb = eng._up(bra if n1 else np.zeros(1, np.uint8))
    tk = eng._up(ket if n2 else np.zeros(1, np.uint8))
    out = eng._alloc(nsb * nsk, zero=True)
    ta = 7
    dd = np.zeros(_cdiv(nsb, ta), nat.pf_desc)
    for j in range(len(dd)):
        dd[j] = (dN[1], dsc[1], tb.data_ptr(), tk.data_ptr(), out.data_ptr(), nn, nn, n1, n2, nsb, nsk, j * ta,
                 min(nsb, (j + 1) * ta))
    m = n1 + n2
    G = 8 if m <= 8 else 16 if m <= 16 else 32
    a16 = lambda x: (x + 15) & ~15  # noqa: E731
    lds = a16(nn * nn * eng.elem) + a16(nsk * n2) + a16(ta * n1) + (256 // G) * 2 * max(m, 1) * eng.elem + 16
    t = eng._up(dd)
    nat.check(eng.lib.tmf_pf_gather_batched(eng.dtype, m, t.data_ptr(), len(dd), lds, eng.stream), "pf")
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(nsb, nsk)
    ref = np.empty((nsb, nsk), N.dtype)
    for a in range(nsb):
        for b in range(nsk):
            ix = np.concatenate((ket[b], bra[a])).astype(int)
            ref[a, b] = scale[0] * (porc.pfaffian(N[np.ix_(ix, ix)]) if m else 1.0)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    if m:  # Pf^2 = det on the GPU values as well
        ix = np.concatenate((ket[0], bra[0])).astype(int)
        np.testing.assert_allclose((got[0, 0] / scale[0]) ** 2, np.linalg.det(N[np.ix_(ix, ix)]), rtol=1e-8)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("kind", ["A", "F"])
def test_nested_products_match_per_cut_gemm(eng, cplx, kind):
    """tmf_nested_products_batched (running sums over nested blocks) against one NumPy product per
    cut: both sides, ragged column counts, cuts at the matrix ends, a gap in the cut list."""
    setup(eng, cplx)
    rng = np.random.default_rng(77)
    D, maxc = 150, 37
    C = rnd(rng, (D, D), cplx)
    Om = rnd(rng, (D, maxc), cplx)
    dC, dOm = dev(eng, C), dev(eng, Om)
    xs = [x for x in range(0, D + 1) if x not in (40, 41, 42)]
    x = np.array(xs + xs)
    side = np.array([0] * len(xs) + [1] * len(xs))
    n = np.where(side == 0, x, D - x)
    ncol = rng.integers(0, maxc + 1, size=len(x))
    ncol[n == 0] = 0
    if kind == "F":
        ncol[(D - n) == 0] = 0
    ld = np.maximum(n, 1) + rng.integers(0, 3, size=len(x))
    sizes = ld * np.maximum(ncol, 1)
    offs = np.concatenate(([0], np.cumsum(sizes)))
    out = eng._alloc(int(offs[-1]), zero=True)
    dest = out.data_ptr() + offs[:-1] * eng.elem
    eng.nested_products(kind, D, dC[1], dOm[1], D, x, side, dest, ncol, ld)
    torch.cuda.synchronize()
    h = out.cpu().numpy()
    for i in range(len(x)):
        if ncol[i] == 0:
            continue
        xi, ni = int(x[i]), int(n[i])
        rows = slice(0, xi) if side[i] == 0 else slice(xi, D)
        if kind == "A":
            cols = rows
        else:
            cols = slice(xi, D) if side[i] == 0 else slice(0, xi)
        ref = C[rows, cols] @ Om[cols, : ncol[i]]
        got = h[offs[i]: offs[i] + ld[i] * ncol[i]].reshape(ncol[i], ld[i]).T[:ni]
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()), err_msg=f"cut {xi} side {side[i]}")


@pytest.mark.parametrize("cplx", [True, False])
def test_recon_error_kernel(eng, cplx):
    """tmf_recon_error_batched: max |T - X diag(w) Y^H| (with and without column reversal of Y),
    max |1 - X^H Y|, ragged sizes across tile borders, and NaN reported as +inf."""
    setup(eng, cplx)
    rng = np.random.default_rng(5)
    items, refs, keep = [], [], []
    for rows, cols, q, rev in [(70, 130, 9, 0), (64, 64, 64, 1), (1, 3, 2, 1), (129, 65, 33, 0)]:
        X, Y, T = rnd(rng, (rows, q), cplx), rnd(rng, (cols, q), cplx), rnd(rng, (rows, cols), cplx)
        w = rng.standard_normal(q)
        dX, dY, dT = dev(eng, X), dev(eng, Y), dev(eng, T)
        keep += [dX, dY, dT]
        Yu = Y[:, ::-1] if rev else Y
        refs.append(np.abs(T - (X * w) @ Yu.conj().T).max())
        items.append(dict(T=dT[1], X=dX[1], Y=dY[1], w=w, rows=rows, cols=cols, q=q, ldt=rows, ldx=rows, ldy=cols,
                          mode=0, y_reverse=rev))
    for inner, r_, c_ in [(100, 70, 70), (5, 3, 66)]:
        X, Y = rnd(rng, (inner, r_), cplx), rnd(rng, (inner, c_), cplx)
        dX, dY = dev(eng, X), dev(eng, Y)
        keep += [dX, dY]
        refs.append(np.abs(np.eye(r_, c_) - X.conj().T @ Y).max())
        items.append(dict(T=0, X=dX[1], Y=dY[1], w=None, rows=r_, cols=c_, q=0, inner=inner, ldx=inner, ldy=inner, mode=1))
    Xn = rnd(rng, (10, 4), cplx)
    Xn[3, 2] = np.nan
    dXn, dTn = dev(eng, Xn), dev(eng, rnd(rng, (10, 10), cplx))
    items.append(dict(T=dTn[1], X=dXn[1], Y=dXn[1], w=None, rows=10, cols=10, q=4, ldt=10, ldx=10, ldy=10, mode=0))
    got = eng.recon_errors(items).cpu().numpy()
    np.testing.assert_allclose(got[:-1], refs, rtol=1e-12)
    assert np.isinf(got[-1])


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("left_only", [True, False])
def test_jacobi_block_variant_beyond_64(eng, cplx, left_only):
    """tmf_jacobi_block_batched (p > 64: X and V in global memory, column blocks staged in LDS) against
    NumPy: singular values, orthonormal vectors, reconstruction; mixed sizes in one launch."""
    setup(eng, cplx)
    rng = np.random.default_rng(21)
    ps = [65, 96, 7, 130]
    Xs = []
    for p in ps:
        if left_only:   # graded lower-triangular factor, as in the sweep
            sv = np.logspace(-0.3, -12, p)
            B = (np.linalg.qr(rnd(rng, (2 * p, p), cplx))[0] * sv) @ np.linalg.qr(rnd(rng, (p, p), cplx))[0]
            Xs.append(np.ascontiguousarray(np.linalg.qr(B)[1].conj().T))
        else:           # Hermitian positive semi-definite, eigenvalues in [0, 1]
            Q = np.linalg.qr(rnd(rng, (p, p), cplx))[0]
            Xs.append((Q * rng.random(p)) @ Q.conj().T)
    dX = [dev(eng, x) for x in Xs]
    dO = [dev(eng, np.zeros_like(x)) for x in Xs]
    ds = [torch.zeros(p, dtype=torch.float64, device="cuda:0") for p in ps]
    dc = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    thr2 = 1e-18 if left_only else 0.0
    eng.jacobi([d[1] for d in dX], [d[1] for d in dO], [s.data_ptr() for s in ds],
               dc.data_ptr() + 4 * np.arange(len(ps)), thr2, ps, ps, ps, left_only=left_only)
    torch.cuda.synchronize()
    for p, X, do, s_ in zip(ps, Xs, dO, ds):
        Out, s = back(do[0], (p, p)), s_.cpu().numpy()
        sref = np.linalg.svd(X, compute_uv=False)
        if left_only:
            c = int(np.sum(sref**2 >= thr2))
            np.testing.assert_allclose(s[:c], sref[:c], rtol=1e-8)
            U = Out[:, :c]
            np.testing.assert_allclose(U.conj().T @ U, np.eye(c), atol=1e-11)
            Ur = np.linalg.svd(X)[0][:, :c]
            np.testing.assert_allclose(np.abs(np.einsum("ij,ij->j", Ur.conj(), U)), 1.0, atol=1e-8)
        else:
            np.testing.assert_allclose(s, sref, rtol=0, atol=1e-13)
            np.testing.assert_allclose(Out.conj().T @ Out, np.eye(p), atol=1e-12)
            np.testing.assert_allclose((Out * s) @ Out.conj().T, X, atol=1e-12)


@pytest.mark.parametrize("cplx", [True, False])
def test_gemm_tall_batched(eng, cplx):
    """tmf_gemm_tall_batched (C = alpha A^H B + beta C, N <= 16, long K; 16 x 16 tiles, 64 rows per step)
    against NumPy, ragged shapes incl. K not a multiple of 64 and M not a multiple of 16."""
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(31)
    shapes = [(56, 8, 768), (1, 1, 1), (17, 16, 65), (290, 16, 300), (33, 5, 64), (16, 16, 1023)]
    for alpha, beta in ((1.0, 0.0), (-0.5, 1.0)):
        As = [rnd(rng, (K, M), cplx) for M, N, K in shapes]
        Bs = [rnd(rng, (K, N), cplx) for M, N, K in shapes]
        Cs = [rnd(rng, (M, N), cplx) for M, N, K in shapes]
        dA, dB, dC = [dev(eng, a) for a in As], [dev(eng, b) for b in Bs], [dev(eng, c) for c in Cs]
        d = np.zeros(len(shapes), nat.gemm_desc)
        tiles = []
        for i, (M, N, K) in enumerate(shapes):
            d[i] = (dA[i][1], dB[i][1], dC[i][1], M, N, K, K, K, M)
            tiles += [(i, t, 0, 0) for t in range(_cdiv(M, 16))]
        t_d, t_t = eng._up(d), eng._up(np.array(tiles, np.int32))
        nat.check(eng.lib.tmf_gemm_tall_batched(eng.dtype, alpha, beta, t_d.data_ptr(), t_t.data_ptr(), len(tiles),
                                                eng.stream), "tall")
        torch.cuda.synchronize()
        for (M, N, K), A, B, C0, dc in zip(shapes, As, Bs, Cs, dC):
            ref = alpha * (A.conj().T @ B) + beta * C0
            np.testing.assert_allclose(back(dc[0], (M, N)), ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()) * K**0.5)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("n,sb,sk,mode", [(1, 6, 5, "near"), (2, 9, 9, "near"), (5, 12, 12, "near"), (8, 15, 20, "near"),
                                          (12, 19, 30, "near"), (13, 20, 31, "far"), (16, 23, 40, "near"),
                                          (16, 64, 64, "far"), (12, 19, 30, "rankdef"), (6, 13, 14, "graded"),
                                          (12, 30, 40, "mixed"), (7, 20, 30, "mixed"), (3, 3, 3, "near"),
                                          (19, 30, 45, "near"), (24, 40, 60, "far"), (32, 64, 64, "near")])
def test_det_ppt_matches_numpy(eng, cplx, n, sb, sk, mode):
    """tmf_det_ppt_batched (one pivoted exchange of the sector matrix per workgroup, every minor a small
    determinant of the exchanged matrix) against numpy.linalg.det.  'near': bra AND ket sets differ from
    the leading ones by 0-3 exchanged orbitals each (the situation of a sweep, d = 0..6: closed forms and
    the queued path); 'far': random sets (d up to 2 n, the 16/32-lane path); 'rankdef': the sector matrix
    has rank < n (every minor is 0); 'graded': rows scaled over 12 decades; 'mixed': 150 ket sets."""
    setup(eng, cplx)
    nat = eng.nat
    rng = np.random.default_rng(300 + n + sk)
    S = rnd(rng, (sb, sk), cplx)
    if mode == "graded":
        S *= np.logspace(0, -12, sb)[:, None]
    if mode == "rankdef":
        S = rnd(rng, (sb, n - 1), cplx) @ rnd(rng, (n - 1, sk), cplx)
    nsb, nsk = (23, 41) if mode != "mixed" else (9, 150)

    def family(size, base, count):
        out = []
        for _ in range(count):
            if mode == "far":
                out.append(np.sort(rng.choice(size, n, replace=False)))
                continue
            k = base.copy()
            for _ in range(rng.integers(0, min(6 if mode == "mixed" else 3, n, size - n) + 1)):
                free = np.setdiff1d(np.arange(size), k)
                if len(free):
                    k[rng.integers(n)] = rng.choice(free)
            out.append(np.sort(k))
        out[0] = base
        return np.stack(out).astype(np.uint8)

    bra = family(sb, np.sort(rng.choice(sb, n, replace=False)), nsb)
    ket = family(sk, np.sort(rng.choice(sk, n, replace=False)), nsk)
    scale = rnd(rng, (1,), cplx)
    dS_, dsc = dev(eng, S), dev(eng, scale)
    tb, tk = eng._up(bra), eng._up(ket)
    out = eng._alloc(nsb * nsk, zero=True)
    ta = 8
    dd = np.zeros(_cdiv(nsb, ta), nat.det_desc)
    for j in range(len(dd)):
        dd[j] = (dS_[1], dsc[1], tb.data_ptr(), tk.data_ptr(), out.data_ptr(), sb, sk, sb, n, nsb, nsk, j * ta,
                 min(nsb, (j + 1) * ta))
    lds = int(nat.ppt_det_lds(eng.elem, sb, sk, nsk, ta, n))
    t = eng._up(dd)
    # (sectors of at most 32 x 32 through the 32-bit-mask kernel, as the sweep launches them)
    nat.check(eng.lib.tmf_det_ppt_batched_w(eng.dtype, t.data_ptr(), len(dd), lds, 32 if max(sb, sk) <= 32 else 64, eng.stream), "ppt")
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(nsb, nsk)
    ref = np.empty((nsb, nsk), S.dtype)
    for a in range(nsb):
        for b in range(nsk):
            ref[a, b] = scale[0] * np.linalg.det(S[np.ix_(bra[a], ket[b])])
    if mode == "rankdef":
        assert np.abs(got).max() <= 1e-9 * np.abs(S).max() ** n
    elif mode == "graded":  # every row of `ref` has its own scale
        for a in range(nsb):
            np.testing.assert_allclose(got[a], ref[a], rtol=0, atol=1e-9 * np.abs(ref[a]).max())
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10 * np.abs(ref).max())


@pytest.mark.parametrize("cplx", [True, False])
def test_house_qr_any_rank(eng, cplx):
    """tmf_house_qr_batched: Q orthonormal to eps and Q R = A for full-rank, graded, exactly rank-deficient,
    wide (m < n), zero and single-column matrices; R or R^H on request (fp64 reference: A itself)."""
    from temfpy_amd import _native as nat

    setup(eng, cplx)
    rng = np.random.default_rng(21)
    shapes = [(40, 40), (98, 55), (260, 130), (17, 1), (5, 9), (1, 1), (64, 33), (300, 70), (33, 20)]
    mats = []
    for i, (m, n) in enumerate(shapes):
        A = rnd(rng, (m, n), cplx)
        if i == 1:    # numerically rank 27 with graded columns (the case that broke Gram-Schmidt)
            A = (rnd(rng, (m, 27), cplx) * np.logspace(0, -7, 27)) @ rnd(rng, (27, n), cplx) * np.logspace(0, -13, n)
        if i == 2:    # exactly rank deficient: duplicated and zero columns
            A[:, 60:100] = A[:, :40]
            A[:, 100:] = 0.0
        if i == 8:
            A[:] = 0.0
        mats.append(A)
    for flag in (0, 1):
        dA = [dev(eng, a) for a in mats]
        dR = [dev(eng, np.full((n, n), 7.0, mats[0].dtype)) for (m, n) in shapes]
        d = np.zeros(len(shapes), nat.qr_desc)
        for i, (m, n) in enumerate(shapes):
            d[i] = (dA[i][1], dR[i][1], m, n, m, n, flag, 0)
        t = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to("cuda:0")
        nat.check(eng.lib.tmf_house_qr_batched(eng.dtype, t.data_ptr(), len(shapes), 300, 130, eng.stream), "qr")
        torch.cuda.synchronize()
        for (m, n), A, da, dr in zip(shapes, mats, dA, dR):
            Q, R = back(da[0], (m, n)), back(dr[0], (n, n))
            if flag:
                R = R.conj().T
            K = min(m, n)
            np.testing.assert_allclose(Q[:, :K].conj().T @ Q[:, :K], np.eye(K), rtol=0, atol=1e-13)
            assert np.all(Q[:, K:] == 0) and np.all(R[K:] == 0) and np.all(np.tril(R, -1) == 0)
            scale = max(np.abs(A).max(), 1e-300)
            np.testing.assert_allclose(Q @ R, A, rtol=0, atol=2e-14 * scale * max(m, n) ** 0.5)
    # flags & 2: R only (|R| is fixed by A^H A = R^H R up to the signs of its rows)
    dA = [dev(eng, a) for a in mats]
    dR = [dev(eng, np.zeros((n, n), mats[0].dtype)) for (m, n) in shapes]
    d = np.zeros(len(shapes), nat.qr_desc)
    for i, (m, n) in enumerate(shapes):
        d[i] = (dA[i][1], dR[i][1], m, n, m, n, 2, 0)
    t = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    nat.check(eng.lib.tmf_house_qr_batched(eng.dtype, t.data_ptr(), len(shapes), 300, 130, eng.stream), "qr")
    torch.cuda.synchronize()
    for (m, n), A, dr in zip(shapes, mats, dR):
        R = back(dr[0], (n, n))
        scale = max(np.abs(A).max(), 1e-300) ** 2
        np.testing.assert_allclose(R.conj().T @ R, A.conj().T @ A, rtol=0, atol=1e-13 * scale * max(m, n))


@pytest.mark.parametrize("cplx", [True, False])
def test_jacobi_compact_rank_deficient(eng, cplx):
    """tmf_jacobi_compact_batched against numpy.linalg.svd: singular values (relative 1e-10 above the
    threshold), orthogonal right vectors reproducing X V = U S, zero columns below the threshold; orders beyond
    the plain LDS kernel (real 130, complex 96 with half of the columns at rounding level), a full-rank matrix
    that does not fit the LDS (global-memory path), a sector of 640 states (round 2 stopped at 512; Gutzwiller at chi = 4096
    reached 635) and tiny problems."""
    from temfpy_amd import _native as nat

    setup(eng, cplx)
    rng = np.random.default_rng(31)
    thr2 = 1e-24
    Xs = []
    for p, r in [(130, 64), (96, 40), (150, 150), (1, 1), (2, 1), (33, 33), (60, 0), (640, 260)]:   # (640: beyond 512 columns, more active ones than threads)
        if r == 0:
            X = np.zeros((p, p), complex if cplx else float)
        else:
            sv = np.logspace(0, -11, r) if r > 2 else np.ones(r)
            X = (np.linalg.qr(rnd(rng, (p, r), cplx))[0] * sv) @ np.linalg.qr(rnd(rng, (p, r), cplx))[0].conj().T
            X = X + 1e-19 * rnd(rng, (p, p), cplx)        # rounding-level columns everywhere
            if r < p:      # lower-triangular like R^H: leading columns carry the weight
                X = np.linalg.qr(X.conj().T)[1].conj().T
        Xs.append(X)
    ps = [len(x) for x in Xs]
    dX = [dev(eng, x) for x in Xs]
    dW = [dev(eng, np.zeros_like(x)) for x in Xs]
    dU = [dev(eng, np.full_like(x, 3.0)) for x in Xs]
    ds = [torch.zeros(p, dtype=torch.float64, device="cuda:0") for p in ps]
    dc = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    d = np.zeros(len(ps), nat.jacobi_desc)
    for i, p in enumerate(ps):
        d[i] = (dX[i][1], dW[i][1], dU[i][1], ds[i].data_ptr(), dc.data_ptr() + 4 * i, thr2, p, p, p, p)
    t = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    nat.check(eng.lib.tmf_jacobi_compact_batched(eng.dtype, t.data_ptr(), len(ps), max(ps), None, eng.stream), "jc")
    torch.cuda.synchronize()
    cnt = dc.cpu().numpy()
    for p, X, du, s_, c in zip(ps, Xs, dU, ds, cnt):
        V, s = back(du[0], (p, p)), s_.cpu().numpy()
        sref = np.linalg.svd(X, compute_uv=False)
        k = int(np.sum(sref**2 >= thr2))
        assert c == k
        np.testing.assert_allclose(s[:k], sref[:k], rtol=1e-10, atol=1e-16)
        np.testing.assert_allclose(V[:, :k].conj().T @ V[:, :k], np.eye(k), rtol=0, atol=1e-13)
        assert np.all(V[:, k:] == 0)
        XV = X @ V[:, :k]
        np.testing.assert_allclose(np.linalg.norm(XV, axis=0), s[:k], rtol=1e-10, atol=1e-16)
        G = XV.conj().T @ XV                                    # columns of X V are orthogonal (= U S)
        np.testing.assert_allclose(G - np.diag(np.diag(G)), 0, atol=1e-14 * max(1.0, sref[0] ** 2))


@pytest.mark.parametrize("cplx", [True, False])
def test_jacobi_compact_left_vectors_of_preconditioned_factor(eng, cplx):
    """desc.V == 0: left singular vectors without accumulator.  Input as in the canonicalisation sweep: the
    conjugate transpose of the twice-QR'd factor of a graded, rank-deficient matrix; the normalised columns must
    be orthonormal to 1e-13 even for singular values of 1e-11 and diagonalise the factor."""
    from temfpy_amd import _native as nat

    setup(eng, cplx)
    rng = np.random.default_rng(41)
    Ws, Rs = [], []
    for p, r in [(141, 110), (64, 64), (20, 7)]:
        sv = np.logspace(0, -11, r)
        N = (np.linalg.qr(rnd(rng, (p, r), cplx))[0] * sv) @ np.linalg.qr(rnd(rng, (2 * p, r), cplx))[0].conj().T   # p x 2p
        R = np.linalg.qr(N.conj().T)[1]                    # N^H = Q R
        R3 = np.linalg.qr(R.conj().T)[1]                   # R^H = Q3 R3
        Rs.append(R3)
        Ws.append(R3.conj().T)
    ps = [len(w) for w in Ws]
    dX = [dev(eng, w) for w in Ws]
    dU = [dev(eng, np.zeros_like(w)) for w in Ws]
    ds = [torch.zeros(p, dtype=torch.float64, device="cuda:0") for p in ps]
    dc = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    d = np.zeros(len(ps), nat.jacobi_desc)
    for i, p in enumerate(ps):
        d[i] = (dX[i][1], 0, dU[i][1], ds[i].data_ptr(), dc.data_ptr() + 4 * i, 1e-24, p, p, p, p)
    t = torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    sw = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    nat.check(eng.lib.tmf_jacobi_compact_batched(eng.dtype, t.data_ptr(), len(ps), max(ps), sw.data_ptr(), eng.stream), "jc")
    torch.cuda.synchronize()
    for p, R3, du, s_, c in zip(ps, Rs, dU, ds, dc.cpu().numpy()):
        V, s = back(du[0], (p, p)), s_.cpu().numpy()
        sref = np.linalg.svd(R3, compute_uv=False)
        k = int(np.sum(sref**2 >= 1e-24))
        assert c == k
        np.testing.assert_allclose(s[:k], sref[:k], rtol=1e-9, atol=1e-17)
        np.testing.assert_allclose(V[:, :k].conj().T @ V[:, :k], np.eye(k), rtol=0, atol=1e-13)
        G = (R3 @ V[:, :k]).conj().T @ (R3 @ V[:, :k])       # V = right singular vectors of R3
        np.testing.assert_allclose(G - np.diag(np.diag(G)), 0, atol=1e-14)
    assert sw.cpu().numpy().max() <= 8                      # preconditioned: few sweeps


def test_utils_block_svd_and_pfaffian_parity():
    """utils.block_svd (utils.py:19-96) and pfaffian.parity (pfaffian.py:396-456) with their small SVDs on the GPU,
    against the reference semantics evaluated with NumPy."""
    from temfpy_amd import pfaffian, utils

    rng = np.random.default_rng(5)
    n, m, k = 12, 9, 6
    U0 = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))[0]
    V0 = np.linalg.qr(rng.normal(size=(m, m)) + 1j * rng.normal(size=(m, m)))[0]
    sv = np.array([0.9, 0.5, 0.5, 0.5, 0.2, 0.2])
    e = np.array([0.7, 0.4, 0.4, 0.4, 0.1, 0.1])          # degenerate groups 1 + 3 + 2
    CLR = (U0[:, :k] * sv) @ V0[:, :k].conj().T
    mix3 = np.linalg.qr(rng.normal(size=(3, 3)) + 1j * rng.normal(size=(3, 3)))[0]
    mix2 = np.linalg.qr(rng.normal(size=(2, 2)) + 1j * rng.normal(size=(2, 2)))[0]
    vL, vR = U0[:, :k].copy(), V0[:, :k].copy()
    vL[:, 1:4], vR[:, 1:4] = vL[:, 1:4] @ mix3, vR[:, 1:4] @ np.linalg.qr(rng.normal(size=(3, 3)))[0]
    vL[:, 4:6] = vL[:, 4:6] @ mix2
    a, b = utils.block_svd(CLR, vL, vR, e, overwrite=False)
    S = a.conj().T @ CLR @ b
    np.testing.assert_allclose(S, np.diag(np.diag(S)), atol=1e-13)
    np.testing.assert_allclose(np.sort(np.abs(np.diag(S)))[::-1], np.sort(sv)[::-1], atol=1e-13)
    assert np.all(np.diag(S).real > 0) and np.abs(np.diag(S).imag).max() < 1e-13
    np.testing.assert_allclose(a.conj().T @ a, np.eye(k), atol=1e-13)
    assert vL is not a and np.abs(vL - U0[:, :k]).max() > 1e-3          # overwrite=False left the input alone
    a2, b2 = utils.block_svd(CLR, vL, vR, e)
    assert a2 is vL and b2 is vR                                         # in place by default (utils.py:65-67)
    # parity: singular values 1 (filled), pairs in (0, 1), zeros
    for n_one, pairs, zeros in [(1, [0.6], 2), (2, [0.8, 0.3], 1), (0, [0.7], 3), (3, [], 2)]:
        s = np.array([1.0] * n_one + [x for x in pairs for _ in range(2)] + [0.0] * zeros)
        p = len(s)
        A = np.linalg.qr(rng.normal(size=(p, p)) + 1j * rng.normal(size=(p, p)))[0]
        B = np.linalg.qr(rng.normal(size=(p, p)) + 1j * rng.normal(size=(p, p)))[0]
        assert pfaffian.parity((A * s) @ B) == n_one % 2
    assert pfaffian.parity(np.zeros((0, 0))) == 0 and pfaffian.parity(np.array([[1j]])) == 1
    assert pfaffian.parity(np.diag([1.0, 0.0])) == 1 and pfaffian.parity(0.5 * np.eye(2)) == 0
    with pytest.raises(RuntimeError):
        pfaffian.parity(np.array([[0.5]]))


@pytest.mark.parametrize("cplx", [True, False])
def test_house_slab_qr(eng, cplx):
    """tmf_house_slab_batched (panel in LDS): in-place thin Q orthonormal to 1e-13 with the column space of A, for
    tall slabs of the range-finder sizes, graded / exactly rank-deficient / zero slabs and n < c."""
    setup(eng, cplx)
    rng = np.random.default_rng(51)
    shapes = [(512, 64), (1023, 64), (300, 64), (64, 64), (40, 64), (1, 64), (777, 33), (130, 1), (257, 17), (96, 64)]
    mats = []
    for i, (n, c) in enumerate(shapes):
        A = rnd(rng, (n, c), cplx)
        if i == 0:      # numerically rank 35, singular values down to 1e-17 (a range-finder slab)
            A = (rnd(rng, (n, 35), cplx) * np.logspace(0, -17, 35)) @ rnd(rng, (35, c), cplx)
        if i == 2:
            A[:, 40:] = 0.0
        if i == 9:
            A[:] = 0.0
        mats.append(A)
    dA = [dev(eng, a) for a in mats]
    eng.house_slab([d[1] for d in dA], [s[0] for s in shapes], [s[0] for s in shapes], [s[1] for s in shapes])
    torch.cuda.synchronize()
    for (n, c), A, da in zip(shapes, mats, dA):
        Q = back(da[0], (n, c))
        K = min(n, c)
        np.testing.assert_allclose(Q[:, :K].conj().T @ Q[:, :K], np.eye(K), rtol=0, atol=1e-13)
        assert np.all(Q[:, K:] == 0)
        scale = max(np.abs(A).max(), 1e-300)
        np.testing.assert_allclose(Q @ (Q.conj().T @ A), A, rtol=0, atol=1e-13 * scale * n ** 0.5)   # span(Q) contains A


@pytest.mark.parametrize("shape", [(220, 110), (256, 128), (100, 40), (128, 128), (250, 112), (64, 90), (7, 3), (130, 0),
                                   (320, 160), (300, 200), (320, 81), (257, 80), (500, 150), (512, 33), (90, 300)])
def test_house_qr_with_all_columns_in_registers(eng, shape):
    """tmf_house_qr_regs_batched (house_reg_kernel: every column of a real block of at most 256 x 128 in registers, the form
    the Gutzwiller canonicalisation sweeps use; house_regp_kernel: panels of 80 / 32 columns in registers for blocks of up to
    320 / 512 rows and any number of columns): thin Q into the scratch, R, against the defining properties; mixed launches
    (several blocks of different sizes, one of them rank deficient, one R-only)."""
    setup(eng, False)
    nat, lib = eng.nat, eng.lib
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    n, c = shape
    mats = [rnd(rng, (n, c), False), rnd(rng, (max(n // 2, 1), max(c // 2, 1)), False)]
    if c >= 8:
        mats.append(rnd(rng, (n, 5), False) @ rnd(rng, (5, c), False))        # rank 5
    dA, dQ, dR, d = [], [], [], np.zeros(len(mats), nat.slab_desc)
    for i, A in enumerate(mats):
        m_, c_ = A.shape
        dA.append(torch.from_numpy(np.ascontiguousarray(A.T).reshape(-1).copy()).to("cuda:0"))
        dQ.append(torch.zeros(m_ * c_ + 2, dtype=torch.float64, device="cuda:0"))
        dR.append(torch.zeros(c_ * c_ + 2, dtype=torch.float64, device="cuda:0"))
        d[i] = (dA[i].data_ptr(), dQ[i].data_ptr(), dR[i].data_ptr(), m_, c_, m_, m_, c_, 4 if i == 1 else 2)    # block 1: R only
    dd = torch.from_numpy(d.view(np.uint8).copy()).to("cuda:0")
    nat.check(lib.tmf_house_qr_regs_batched(nat.TMF_F64, dd.data_ptr(), len(mats), max(a.shape[0] for a in mats),
                                            max(a.shape[1] for a in mats), eng.stream), "tmf_house_qr_regs_batched")
    torch.cuda.synchronize()
    for i, A in enumerate(mats):
        m_, c_ = A.shape
        if c_ == 0:
            continue
        R = dR[i].cpu().numpy()[: c_ * c_].reshape(c_, c_).T
        scale = max(np.abs(A).max(), 1e-300)
        assert np.isfinite(R).all() and np.allclose(np.tril(R, -1), 0)
        np.testing.assert_allclose(R.T @ R, A.T @ A, rtol=0, atol=1e-13 * scale ** 2 * max(m_, c_))
        if i == 1:
            continue
        Q = dQ[i].cpu().numpy()[: m_ * c_].reshape(c_, m_).T
        K = min(m_, c_)
        np.testing.assert_allclose(Q[:, :K].T @ Q[:, :K], np.eye(K), rtol=0, atol=1e-13)
        np.testing.assert_allclose(Q[:, :K] @ R[:K], A, rtol=0, atol=1e-13 * scale * max(m_, c_))


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("shape", [(320, 160), (220, 110), (100, 130), (64, 64), (9, 4)])
def test_house_qr_with_q_formed_later(eng, shape, cplx):
    """flags & 8 of both slab entry points + tmf_house_form_q_batched: the factorisation leaves the reflectors in A and their
    scalars in the caller's buffer, Q is formed in place by a later launch (the Gutzwiller sweeps form every Q of a sweep in
    one launch after the chain of factorisations).  Against the one-launch form of the panel kernel: R and Q bit for bit
    (same arithmetic per column); from the register kernel: the defining properties."""
    setup(eng, cplx)
    nat, lib = eng.nat, eng.lib
    dt, tdt = (nat.TMF_C128, torch.complex128) if cplx else (nat.TMF_F64, torch.float64)
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + cplx)
    n, c = shape
    mats = [rnd(rng, (n, c), cplx), rnd(rng, (max(n // 3, 1), max(c // 2, 1)), cplx)]
    if c >= 8:
        mats.append(rnd(rng, (n, 5), cplx) @ rnd(rng, (5, c), cplx))
    mx_n, mx_c = max(a.shape[0] for a in mats), max(a.shape[1] for a in mats)

    def run(entry, deferred):
        dA, dQ, dR, d = [], [], [], np.zeros(len(mats), nat.slab_desc)
        for i, A in enumerate(mats):
            m_, c_ = A.shape
            dA.append(torch.from_numpy(np.ascontiguousarray(A.T).reshape(-1).copy()).to("cuda:0"))
            dQ.append(torch.zeros(m_ * c_ + 2, dtype=tdt, device="cuda:0"))
            dR.append(torch.zeros(c_ * c_ + 2, dtype=tdt, device="cuda:0"))
            d[i] = (dA[i].data_ptr(), dQ[i].data_ptr(), dR[i].data_ptr(), m_, c_, m_, m_, c_, 8 if deferred else 0)
        dd = torch.from_numpy(d.view(np.uint8).copy()).to("cuda:0")
        nat.check(getattr(lib, entry)(dt, dd.data_ptr(), len(mats), mx_n, mx_c, eng.stream), entry)
        if deferred:
            torch.cuda.synchronize()
            for i, A in enumerate(mats):      # the scalars: c entries, zero past min(n, c); nothing written behind them
                t = dQ[i].cpu().numpy()
                K = min(A.shape)
                assert np.all(t[K:] == 0)
            nat.check(lib.tmf_house_form_q_batched(dt, dd.data_ptr(), len(mats), mx_n, mx_c, eng.stream), "tmf_house_form_q_batched")
        torch.cuda.synchronize()
        Qs = [dA[i].cpu().numpy()[: A.shape[0] * A.shape[1]].reshape(A.shape[1], A.shape[0]).T for i, A in enumerate(mats)]
        Rs = [dR[i].cpu().numpy()[: A.shape[1] ** 2].reshape(A.shape[1], A.shape[1]).T for i, A in enumerate(mats)]
        return Qs, Rs

    Q0, R0 = run("tmf_house_slab_batched", False)
    Q1, R1 = run("tmf_house_slab_batched", True)
    for a, b in zip(Q0 + R0, Q1 + R1):
        assert np.array_equal(a, b)
    Q2, R2 = run("tmf_house_qr_regs_batched", True)
    Q3, R3 = run("tmf_house_qr_regs_batched", False)           # Q over A in the same launch
    for A, Q, R in list(zip(mats, Q2, R2)) + list(zip(mats, Q3, R3)):
        m_, c_ = A.shape
        K = min(m_, c_)
        scale = np.abs(A).max()
        np.testing.assert_allclose(Q[:, :K].conj().T @ Q[:, :K], np.eye(K), rtol=0, atol=1e-13)
        np.testing.assert_allclose(Q[:, :K] @ R[:K], A, rtol=0, atol=1e-13 * scale * max(m_, c_))


def test_export_words_verdict_and_conditional_launches(eng):
    """Plumbing of the block-local elimination: tmf_export_words (device -> page-locked host memory by a kernel),
    tmf_diag_inverse_verdict (statistics -> device flag + summary in host memory) and tmf_launch_condition (GEMM / gather /
    pivoted-LU launches that return at once while the device flag is clear)."""
    import ctypes

    setup(eng, True)
    nat, lib = eng.nat, eng.lib
    # export
    src = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    dst = torch.zeros(1000, dtype=torch.float64).pin_memory()
    nat.check(lib.tmf_export_words(dst.data_ptr(), src.data_ptr(), 8000, eng.stream), "export")
    torch.cuda.synchronize()
    assert np.array_equal(dst.numpy(), np.arange(1000) * 0.5)
    assert lib.tmf_export_words(dst.data_ptr(), src.data_ptr(), 7, eng.stream) != 0       # not a multiple of 4
    # verdict: (min pivot^2, max inverse^2) per matrix
    stats = np.array([[0.25, 4.0], [1e-6, 9.0], [1.0, 0.0]])
    d_stats = torch.from_numpy(stats.reshape(-1)).cuda()
    d_flag = torch.full((1,), 7, dtype=torch.int32, device="cuda")
    summary = torch.zeros(3, dtype=torch.float64).pin_memory()
    for cap, force, want in ((10.0, 0, 0), (2.5, 0, 1), (10.0, 1, 1)):
        nat.check(lib.tmf_diag_inverse_verdict(d_stats.data_ptr(), 3, cap, force, d_flag.data_ptr(), summary.data_ptr(), eng.stream), "verdict")
        torch.cuda.synchronize()
        assert int(d_flag.item()) == want
        np.testing.assert_allclose(summary.numpy(), [1e-3, 3.0, want])
    d_stats[3] = float("nan")
    nat.check(lib.tmf_diag_inverse_verdict(d_stats.data_ptr(), 3, 10.0, 0, d_flag.data_ptr(), summary.data_ptr(), eng.stream), "verdict")
    torch.cuda.synchronize()
    assert int(d_flag.item()) == 1                                                         # a NaN always triggers the fallback
    # conditional scope around a GEMM: C stays untouched while the flag is 0, is computed when it is 1
    rng = np.random.default_rng(5)
    A, B = rnd(rng, (40, 30), True), rnd(rng, (30, 20), True)
    dA, dB = dev(eng, A), dev(eng, B)
    dC = eng._alloc(40 * 20, zero=True)
    for flag in (0, 1):
        d_flag.fill_(flag)
        lib.tmf_launch_condition(d_flag.data_ptr())
        try:
            eng.gemm(0, 1.0, 0.0, [dA[1]], [dB[1]], [dC.data_ptr()], [40], [20], [30], [40], [30], [40])
        finally:
            lib.tmf_launch_condition(None)
        torch.cuda.synchronize()
        got = back(dC, (40, 20))
        np.testing.assert_allclose(got, A @ B if flag else np.zeros((40, 20)), atol=1e-12)
    eng.gemm(0, 2.0, 0.0, [dA[1]], [dB[1]], [dC.data_ptr()], [40], [20], [30], [40], [30], [40])   # scope ended: runs
    torch.cuda.synchronize()
    np.testing.assert_allclose(back(dC, (40, 20)), 2 * (A @ B), atol=1e-12)


def test_single_call_sweep_entry_point_and_accessors():
    """tmf_slater_sweep (the one-call form of the staged sweep ABI, include/temfpy_hip.h "Sweep level") through ctypes,
    read back with the tmf_result_* accessors, against the Python entry point that drives the staged calls."""
    import ctypes
    from tests_inputs import random_hopping
    from temfpy_amd import slater, _native as nat

    L, chi = 40, 24
    C, _ = slater.correlation_matrix(random_hopping(L, 4))
    ref = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False)
    lib = nat.load()
    ctx, res = ctypes.c_void_p(), ctypes.c_void_p()
    nat.check(lib.tmf_ctx_create(0, ctypes.byref(ctx)), "tmf_ctx_create")
    Cc = np.ascontiguousarray(C, np.complex128)
    par = nat.SweepParams(L=L, chi_max=chi, svd_min=1e-6, degeneracy_tol=1e-12, sectors=None, ortho_center=L // 2, site_lo=0,
                          site_hi=L, n_sectors=0, is_complex=1, host_threads=4, flags=nat.SWEEP_CHECKS)
    nat.check(lib.tmf_slater_sweep(ctx, Cc.ctypes.data, ctypes.byref(par), 0.0, ctypes.byref(res)), "tmf_slater_sweep")
    try:
        for b in range(L + 1):
            v = nat.BondView()
            nat.check(lib.tmf_result_bond(res, b, ctypes.byref(v)), "tmf_result_bond")
            bd = ref.bonds[b]
            assert (v.chi, v.k, v.n_filled_left, v.n_filled_right) == (bd.chi, len(bd.e), bd.n_filled_left, bd.n_filled_right)
            lam = np.ctypeslib.as_array(ctypes.cast(v.lam_raw, ctypes.POINTER(ctypes.c_double)), (v.chi,))
            masks = np.ctypeslib.as_array(ctypes.cast(v.masks, ctypes.POINTER(ctypes.c_uint64)), (v.chi, 2))
            assert np.array_equal(lam, bd.lam_raw) and np.array_equal(masks, bd.masks)
        for i in range(L):
            sv = nat.SiteView()
            nat.check(lib.tmf_result_site(res, i, ctypes.byref(sv)), "tmf_result_site")
            s = ref.sites[i]
            assert (sv.chi_bra, sv.chi_ket, sv.n_blocks) == (s.chi_bra, s.chi_ket, len(s.blocks))
            assert complex(sv.det_always[0], sv.det_always[1]) == s.det_always
            for j, (q, r0, r1, c0, c1, blk) in enumerate(s.blocks):
                bv = nat.BlockView()
                nat.check(lib.tmf_result_block(res, i, j, ctypes.byref(bv)), "tmf_result_block")
                assert (bv.q, bv.r0, bv.r1, bv.c0, bv.c1) == (q, r0, r1, c0, c1)
                got = np.ctypeslib.as_array(ctypes.cast(bv.data, ctypes.POINTER(ctypes.c_double)), (r1 - r0, c1 - c0, 2))
                assert np.array_equal(got[..., 0] + 1j * got[..., 1], blk)
        checks, n = (ctypes.c_double * 8)(), ctypes.c_int32()
        nat.check(lib.tmf_result_checks(res, checks, ctypes.byref(n)), "tmf_result_checks")
        assert n.value == 5 and max(checks[:5]) < 1e-6
        assert lib.tmf_result_bond(res, L + 3, ctypes.byref(nat.BondView())) == -1
    finally:
        lib.tmf_result_free(res)
        lib.tmf_ctx_destroy(ctx)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("left_only", [True, False])
def test_jacobi_every_small_size(eng, cplx, left_only):
    """One-sided Jacobi on every problem size 1 .. 24 and a few larger odd ones (an odd number of active columns leaves
    one player of the tournament idle; sizes up to 16 run with 64 lanes per pair): converges within a few sweeps,
    singular values to 1e-12, orthonormal vectors.  Includes the 3 x 3 triangular factor of a cut of an L = 14 chain that
    did not converge before the idle player's partner was kept out of the reductions (tests/soak/soak_small.py, seed 1919)."""
    setup(eng, cplx)
    rng = np.random.default_rng(77)
    ps = list(range(1, 25)) + [31, 33, 47, 63]
    Xs = [rnd(rng, (p, p), cplx) * np.logspace(0, -3, p)[None, :] for p in ps]
    Xs[2] = np.array([[0.47502447471382064, 0.0, 0.0], [0.05627941268549051, -0.10970793908097089, 0.0],
                      [-0.0913023802829401, 0.05655712241246387, -0.07941423969753784]]).astype(Xs[2].dtype)
    dX = [dev(eng, x) for x in Xs]
    dO = [dev(eng, np.zeros_like(x)) for x in Xs]
    ds = [torch.zeros(p, dtype=torch.float64, device="cuda:0") for p in ps]
    dc = torch.zeros(len(ps), dtype=torch.int32, device="cuda:0")
    eng._keep.clear()
    eng.jacobi([d[1] for d in dX], [d[1] for d in dO], [s.data_ptr() for s in ds], dc.data_ptr() + 4 * np.arange(len(ps)),
               1e-30, ps, ps, ps, left_only=left_only)
    torch.cuda.synchronize()
    sweeps = [t for t in eng._keep if getattr(t, "dtype", None) == torch.int32][-1].cpu().numpy()
    assert sweeps.max() <= 12, sweeps.tolist()
    for p, X, do, s_ in zip(ps, Xs, dO, ds):
        O, s = back(do[0], (p, p)), s_.cpu().numpy()
        np.testing.assert_allclose(s, np.linalg.svd(X, compute_uv=False), rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(O.conj().T @ O, np.eye(p), atol=1e-12)
        if left_only:       # U diag(s) U^H = X X^H
            np.testing.assert_allclose((O * s**2) @ O.conj().T, X @ X.conj().T, atol=1e-12)
        else:               # right vectors: X V has orthogonal columns of norm s
            np.testing.assert_allclose(np.linalg.norm(X @ O, axis=0), s, rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("kernel", ["tmf_house_qr_batched", "tmf_house_slab_batched", "tmf_house_qr_regs_batched"])
def test_householder_far_past_the_rank(eng, kernel):
    """Householder QR of exactly rank-deficient blocks with many more steps than rank (Gutzwiller-projected tensors: zero rows,
    rank 4 of 16 x 19): past the rank every step works on the rounding noise of the previous one, whose squared length
    underflows after a dozen steps; such a column is a zero column.  (Before that guard: tau = 0 / 0, NaN in R, wrong
    Schmidt values from that bond on - tests/soak/soak_gutzwiller.py seed 67; the first matrix is the one from that case.)"""
    lib = eng.lib
    rng = np.random.default_rng(3)
    mats = [np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernels", "qr_rank4_16x19.npy"))]
    for m, n, r, cplx in ((16, 19, 4, True), (40, 24, 3, True), (30, 30, 2, False), (64, 70, 5, False), (12, 40, 1, True)):
        A = rnd(rng, (m, r), cplx) @ rnd(rng, (r, n), cplx)
        A[rng.permutation(m)[: m // 2]] = 0          # exact zero rows
        mats.append(A)
    for V in mats:
        cplx = np.iscomplexobj(V)
        setup(eng, cplx)
        dt, tdt = (eng.nat.TMF_C128, torch.complex128) if cplx else (eng.nat.TMF_F64, torch.float64)
        m, n = V.shape
        dA = torch.from_numpy(np.ascontiguousarray(V.T).reshape(-1).copy()).to("cuda:0")
        dR = torch.zeros(n * n, dtype=tdt, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        if kernel == "tmf_house_qr_batched":
            d = np.zeros(1, eng.nat.qr_desc)
            d[0] = (dA.data_ptr(), dR.data_ptr(), m, n, m, n, 0, 0)
            dd = torch.from_numpy(d.view(np.uint8).copy()).to("cuda:0")
            eng.nat.check(lib.tmf_house_qr_batched(dt, dd.data_ptr(), 1, m, n, st), kernel)
            Q = None
        else:
            dQ = torch.zeros(m * n + 2, dtype=tdt, device="cuda:0")
            d = np.zeros(1, eng.nat.slab_desc)
            d[0] = (dA.data_ptr(), dQ.data_ptr(), dR.data_ptr(), m, n, m, m, n, 0)
            dd = torch.from_numpy(d.view(np.uint8).copy()).to("cuda:0")
            eng.nat.check(getattr(lib, kernel)(dt, dd.data_ptr(), 1, m, n, st), kernel)
        torch.cuda.synchronize()
        R = dR.cpu().numpy().reshape(n, n).T
        Q = dA.cpu().numpy().reshape(n, m).T
        assert np.isfinite(R).all() and np.isfinite(Q).all()
        scale = np.abs(V).max() ** 2
        np.testing.assert_allclose(R.conj().T @ R, V.conj().T @ V, rtol=0, atol=1e-13 * scale * max(m, n))
        k = min(m, n)
        np.testing.assert_allclose(Q[:, :k] @ R[:k], V, rtol=0, atol=1e-13 * np.abs(V).max() * max(m, n))
