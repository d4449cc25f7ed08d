"""CPU oracle for the Gutzwiller projections of a finite fermionic MPS.  TEST INFRASTRUCTURE ONLY.

Restates ``/root/reference/src/temfpy/gutzwiller.py`` (``abrikosov`` :95-281, ``abrikosov_ph`` :284-486;
citations are ``file:line`` in that tree) on plain dense NumPy tensors.  Nothing in ``temfpy_amd`` imports it.

What the reference does on this path is bookkeeping on a TeNPy ``MPS`` followed by TeNPy's
``MPS.canonical_form_finite`` (called at gutzwiller.py:266 / :471).  TeNPy (``physics-tenpy >= 1.1.0``,
pyproject.toml:36) is a third-party dependency that is not installed anywhere we can run, so

    **parity unpinned** against TeNPy's own arithmetic;

the restatement below follows the published algorithm of ``canonical_form_finite`` (a QR sweep to the right,
then an SVD sweep to the left that discards Schmidt values <= cutoff of the normalised state and renormalises)
and is pinned by an independent brute-force check instead (``project_state`` / ``schmidt_values_of_state``:
the projector applied to the full 2^L state vector, Schmidt values by one dense SVD per cut), see
``tests/test_oracle_gutzwiller.py``.  Everything gauge invariant (Schmidt values per bond and charge sector,
the state itself) is therefore fixed; the gauge inside degenerate Schmidt multiplets is not.

Conventions: fermion tensors ``T[i]`` have shape (2, chi_l, chi_r) (p = occupation), the state is
``T[0] .. T[oc-1] diag(lam_c) T[oc] .. T[L-1]`` (forms A..A lam B..B, slater.py:1348);  ``q[b]`` holds the
conserved charge to the left of bond ``b`` for every Schmidt index (particle number, or parity).
Spin tensors have shape (2, chi_l, chi_r) with p = 0: spin down, p = 1: spin up.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------
# projection (gutzwiller.py:217-244 and :399-444)
# --------------------------------------------------------------------------------------
def group_and_project(T, q, lam_c, oc, kind="ph", conserve="N", parity=0, q_left=0):
    """Pairs of sites (2i, 2i+1) -> one spin-1/2 site.

    kind "ph"  (abrikosov_ph): physical states |00> -> down, |11> -> up (``parity_mask(leg_p)``,
        gutzwiller.py:412: even occupation; the grouped leg is ordered 00,01,10,11 so the two kept states
        come out as [down, up]); virtual indices with charge parity ``parity`` (:418-419).
    kind "std" (abrikosov): |10> -> up, |01> -> down (``mask(leg_p, 1)`` :230: occupation 1 resp. odd
        parity; kept states in leg order 01, 10 = [down, up]); virtual indices of bond ``idx`` with charge
        ``q_left + idx`` (:236-238; exactly for conserve="N", modulo 2 for "parity").

    Returns (list of spin tensors (2, chi_l', chi_r'), list of kept index arrays per spin bond).
    The centre Schmidt values are multiplied in, so the product of the returned tensors is the projected,
    unnormalised state (TeNPy's ``group_sites`` keeps the state while regrouping)."""
    L = len(T)
    assert L % 2 == 0, "Odd-length MPS cannot represent an Abrikosov fermion Hilbert space"  # gutzwiller.py:158
    T = [np.asarray(t) for t in T]
    if oc < L:
        T[oc] = T[oc] * np.asarray(lam_c)[None, :, None]
    else:
        T[L - 1] = T[L - 1] * np.asarray(lam_c)[None, None, :]
    keep = []
    for idx in range(L // 2 + 1):
        qq = np.asarray(q[2 * idx])
        if kind == "ph":
            m = qq % 2 == parity % 2
        elif conserve == "N":
            m = qq == q_left + idx
        else:
            m = qq % 2 == (q_left + idx) % 2
        keep.append(np.nonzero(m)[0])
    out = []
    for j in range(L // 2):
        a, b = T[2 * j], T[2 * j + 1]
        pairs = ((0, 0), (1, 1)) if kind == "ph" else ((0, 1), (1, 0))   # [down, up]
        S = np.stack([a[p1] @ b[p2] for p1, p2 in pairs])
        out.append(S[:, keep[j]][:, :, keep[j + 1]])
    return out, keep


def spin_charges(q, keep, kind="ph", conserve="N", offset=0):
    """2 S^z to the left of every kept index of every spin bond: ``number - offset - bond_index``
    (gutzwiller.py:333, :438-441); None when no charge survives (:443-444, :244)."""
    if kind != "ph" or conserve != "N":
        return None
    return [np.asarray(q[2 * j])[k] - offset - j for j, k in enumerate(keep)]


# --------------------------------------------------------------------------------------
# TeNPy's MPS.canonical_form_finite(renormalize=True, cutoff) restated (call sites gutzwiller.py:266, :471)
# --------------------------------------------------------------------------------------
def canonical_form_finite(M, cutoff=1e-12):
    """Right-canonical form of the finite MPS with site tensors ``M[j]`` (2, chi_l, chi_r).

    Sweep 1 (left to right): QR of the (p chi_l) x chi_r matrix, R pushed into the next site.
    Sweep 2 (right to left): SVD of the chi_l x (p chi_r) matrix, ``B = V^h``, ``U S`` pushed to the left;
    Schmidt values <= cutoff (of the normalised state) are discarded and the rest renormalised.
    (No charges: the tensors TeNPy holds after ``drop_charge``, gutzwiller.py:244 / :444.)
    Returns (B list, S list (L+1 entries), norm of the input state)."""
    L = len(M)
    M = [np.array(m, dtype=np.result_type(m.dtype, float)) for m in M]
    # sweep 1
    for j in range(L - 1):
        d, cl, cr = M[j].shape
        Q, R = np.linalg.qr(M[j].reshape(d * cl, cr))
        M[j] = Q.reshape(d, cl, -1)
        M[j + 1] = np.einsum("ab,pbc->pac", R, M[j + 1])
    nrm = np.linalg.norm(M[L - 1])
    M[L - 1] = M[L - 1] / nrm
    S = [None] * (L + 1)
    S[L] = np.ones(1)
    B = [None] * L
    for j in range(L - 1, -1, -1):
        d, cl, cr = M[j].shape
        X = M[j].transpose(1, 0, 2).reshape(cl, d * cr)
        U, s, Vh = np.linalg.svd(X, full_matrices=False)
        k = s > cutoff
        if j == 0:
            k = np.arange(len(s)) < 1
        U, s, Vh = U[:, k], s[k], Vh[k]
        s = s / np.linalg.norm(s)
        B[j] = Vh.reshape(-1, d, cr).transpose(1, 0, 2)
        S[j] = s
        if j > 0:
            M[j - 1] = np.einsum("pab,bc->pac", M[j - 1], U * s[None, :])
    return B, S, nrm


def canonical_form_finite_blocks(M, ql, cutoff=1e-12):
    """Same algorithm with U(1) charge blocks: ``ql[j]`` = charge (2 S^z to the left) of every index of spin
    bond j; p = 0 (down) lowers it by 1, p = 1 (up) raises it by 1 from bond j to bond j+1.
    New Schmidt indices are ordered by charge, inside a charge by decreasing Schmidt value.
    Returns (B, S, charges per bond, norm)."""
    L = len(M)
    M = [np.array(m, dtype=np.result_type(m.dtype, float)) for m in M]
    ql = [np.asarray(x) for x in ql]
    dq = (-1, +1)
    # sweep 1: blockwise QR over the right charge
    for j in range(L - 1):
        d, cl, cr = M[j].shape
        newQ = np.zeros_like(M[j])
        R = np.zeros((cr, cr), M[j].dtype)
        for c in np.unique(ql[j + 1]):
            cols = np.nonzero(ql[j + 1] == c)[0]
            rows = [(p, np.nonzero(ql[j] == c - dq[p])[0]) for p in range(d)]
            X = np.concatenate([M[j][p][np.ix_(r, cols)] for p, r in rows], axis=0)
            if X.shape[0] == 0:
                continue
            Q, Rb = np.linalg.qr(X, mode="reduced")
            # keep the shape: pad Q with zero columns / R with zero rows when rows < cols
            kq = Q.shape[1]
            o = 0
            for p, r in rows:
                newQ[p][np.ix_(r, cols[:kq])] = Q[o:o + len(r)]
                o += len(r)
            R[np.ix_(cols[:kq], cols)] = Rb
        M[j] = newQ
        M[j + 1] = np.einsum("ab,pbc->pac", R, M[j + 1])
    nrm = np.linalg.norm(M[L - 1])
    M[L - 1] = M[L - 1] / nrm
    S, B, Q = [None] * (L + 1), [None] * L, [None] * (L + 1)
    S[L], Q[L] = np.ones(M[L - 1].shape[2]), ql[L]
    for j in range(L - 1, -1, -1):
        d, cl, cr = M[j].shape
        qr_ = Q[j + 1]
        rowsB, s_all, q_all, Us = [], [], [], []
        for c in np.unique(ql[j]):
            r = np.nonzero(ql[j] == c)[0]
            cols = [(p, np.nonzero(qr_ == c + dq[p])[0]) for p in range(d)]
            X = np.concatenate([M[j][p][np.ix_(r, cc)] for p, cc in cols], axis=1)
            if X.shape[1] == 0:
                continue
            U, s, Vh = np.linalg.svd(X, full_matrices=False)
            k = s > cutoff
            U, s, Vh = U[:, k], s[k], Vh[k]
            for i in range(len(s)):
                row = np.zeros((d, cr), M[j].dtype)
                o = 0
                for p, cc in cols:
                    row[p, cc] = Vh[i, o:o + len(cc)]
                    o += len(cc)
                rowsB.append(row)
                u = np.zeros(cl, M[j].dtype)
                u[r] = U[:, i] * s[i]
                Us.append(u)
            s_all += list(s)
            q_all += [c] * len(s)
        s_all = np.array(s_all)
        nn = np.linalg.norm(s_all)
        B[j] = np.stack(rowsB, axis=1) if rowsB else np.zeros((d, 0, cr))
        S[j], Q[j] = s_all / nn, np.array(q_all)
        if j > 0:
            Umat = np.stack(Us, axis=1) / nn
            M[j - 1] = np.einsum("pab,bc->pac", M[j - 1], Umat)
    return B, S, Q, nrm


# --------------------------------------------------------------------------------------
# independent brute-force check (small L): projector on the full state vector
# --------------------------------------------------------------------------------------
def state_vector(T, lam_c=None, oc=None):
    """psi[p_0, .., p_{L-1}] of the MPS (optionally with centre Schmidt values on bond ``oc``)."""
    L = len(T)
    psi = np.ones((1, 1), complex)  # [phys..., bond]
    for i, t in enumerate(T):
        if lam_c is not None and i == oc:
            psi = psi * np.asarray(lam_c)[None, :]
        psi = np.tensordot(psi, t, axes=(1, 1))      # [P, p, r]
        psi = psi.reshape(-1, t.shape[2])
    if lam_c is not None and oc == L:
        psi = psi * np.asarray(lam_c)[None, :]
    assert psi.shape[1] == 1
    return psi[:, 0].reshape((2,) * L)


def project_state(psi, kind="ph"):
    """Gutzwiller projector on the occupation-basis amplitudes: spin amplitudes chi[s_0, .., s_{L/2-1}],
    s = 0 down, 1 up (kind "ph": 00 -> down, 11 -> up; kind "std": 01 -> down, 10 -> up).  No fermionic
    signs arise: TeNPy's ``group_sites`` / ``iproject`` act on the MPS tensors exactly like this."""
    L = psi.ndim
    out = psi
    for j in range(L // 2):
        # axes (j, j+1) of the partially projected array are the next pair
        a = np.moveaxis(out, (j, j + 1), (0, 1))
        pr = np.stack([a[0, 0], a[1, 1]]) if kind == "ph" else np.stack([a[0, 1], a[1, 0]])
        out = np.moveaxis(pr, 0, j)
    return out


def schmidt_values_of_state(chi, cutoff=1e-12):
    """Normalised Schmidt values (descending) at every bond of a spin state chi[s_0..s_{n-1}]."""
    n = chi.ndim
    chi = chi / np.linalg.norm(chi)
    out = [np.ones(1)]
    for b in range(1, n):
        s = np.linalg.svd(chi.reshape(2**b, -1), compute_uv=False)
        s = s[s > cutoff]
        out.append(s / np.linalg.norm(s))
    out.append(np.ones(1))
    return out


# --------------------------------------------------------------------------------------
# infinite MPS (gutzwiller.py:197-206, :230-242 / :392-398, :416-447; canonical form at :272 / :475)
# --------------------------------------------------------------------------------------
def group_and_project_cell(T, q, cell_charge, kind="ph", conserve="N", parity=0, q_left=0):
    """Unit cell of an infinite fermionic MPS in B form: ``T[i]`` (2, chi_i, chi_{i+1}) with chi_L = chi_0, ``q[b]``
    (b < L) the charge labels of bond b.  The labels of the closing bond are those of bond 0 plus ``cell_charge`` (what
    ``mps.gauge_total_charge(qtotal=...)`` arranges, gutzwiller.py:212 / :398), and the masks of bond ``idx`` are those of
    the finite case with ``idx_next = (idx + 1) % L`` (:234-238), i.e. the closing bond keeps the indices bond 0 keeps.
    Returns (spin tensors of the cell, kept index arrays of the L/2 + 1 bonds, first = last)."""
    L = len(T)
    assert L % 2 == 0, "Odd-length MPS cannot represent an Abrikosov fermion Hilbert space"
    keep = []
    for idx in range(L // 2):
        qq = np.asarray(q[2 * idx])
        if kind == "ph":
            m = qq % 2 == parity % 2
        elif conserve == "N":
            m = qq == q_left + idx
        else:
            m = qq % 2 == (q_left + idx) % 2
        keep.append(np.nonzero(m)[0])
    keep.append(keep[0])
    pairs = ((0, 0), (1, 1)) if kind == "ph" else ((0, 1), (1, 0))
    out = []
    for j in range(L // 2):
        a, b = np.asarray(T[2 * j]), np.asarray(T[2 * j + 1])
        S = np.stack([a[p1] @ b[p2] for p1, p2 in pairs])
        out.append(S[:, keep[j]][:, :, keep[j + 1]])
    return out, keep


def canonical_form_infinite(M, eps=1e-15):
    """Right-canonical form of the infinite MPS with unit cell ``M[j]`` (2, chi_j, chi_{j+1}), chi_L = chi_0: the published
    algorithm behind TeNPy's ``MPS.canonical_form_infinite1`` on dense matrices.  Dominant left / right eigenvectors l, r of
    the transfer matrix of the cell on bond 0 (Gram matrices, trace 1), pushed through the cell to every bond; per bond
    r_j = Y Y^H, l_j = X^H X (eigenvalues below ``eps`` dropped), X Y = U S V^H, and B_j = (Y_j V_j)^+ M_j (Y_{j+1} V_{j+1})
    up to the norm.  Returns (B list, S list (L + 1 entries, first = last), eta = dominant eigenvalue = norm^2 per cell)."""
    L = len(M)
    M = [np.asarray(m, dtype=np.result_type(m.dtype, float)) for m in M]
    chi = M[0].shape[1]

    def push_r(j, r):      # right Gram matrix from bond j+1 to bond j
        return sum(M[j][p] @ r @ M[j][p].conj().T for p in range(2))

    def push_l(j, l_):     # left Gram matrix from bond j to bond j+1
        return sum(M[j][p].conj().T @ l_ @ M[j][p] for p in range(2))

    def cell(push, order, x):
        for j in order:
            x = push(j, x)
        return x

    def dominant(push, order):
        E = np.zeros((chi * chi, chi * chi), complex)
        for k in range(chi * chi):
            e = np.zeros(chi * chi, complex)
            e[k] = 1
            E[:, k] = cell(push, order, e.reshape(chi, chi)).reshape(-1)
        w, v = np.linalg.eig(E)
        i = int(np.argmax(np.abs(w)))
        x = v[:, i].reshape(chi, chi)
        x = x / np.trace(x)
        assert np.abs(x - x.conj().T).max() < 1e-8 * np.abs(x).max(), "transfer matrix not injective"
        return w[i], 0.5 * (x + x.conj().T)

    eta_r, r0 = dominant(push_r, range(L - 1, -1, -1))
    eta_l, l0 = dominant(push_l, range(L))
    assert abs(eta_r - eta_l) < 1e-9 * abs(eta_r)
    eta = float(np.real(eta_r))
    ls, rs = [l0], [None] * L + [r0]
    for j in range(L - 1):
        ls.append(push_l(j, ls[-1]))
    for j in range(L - 1, 0, -1):
        rs[j] = push_r(j, rs[j + 1])
    rs[0] = r0
    G, Gi, S = [], [], []
    for j in range(L):
        wr, ur = np.linalg.eigh(rs[j] / np.trace(rs[j]).real)
        k = wr > eps
        Y, Yi = ur[:, k] * np.sqrt(wr[k]), (ur[:, k] / np.sqrt(wr[k])).conj().T
        wl, ul = np.linalg.eigh(ls[j] / np.trace(ls[j]).real)
        k = wl > eps
        X = (ul[:, k] * np.sqrt(wl[k])).conj().T
        U, s, Vh = np.linalg.svd(X @ Y, full_matrices=False)
        k = s > np.sqrt(eps) * np.linalg.norm(s)
        V = Vh.conj().T[:, k]
        G.append(Y @ V)
        Gi.append(V.conj().T @ Yi)
        S.append(s[k] / np.linalg.norm(s[k]))
    G.append(G[0])
    S.append(S[0])
    B = []
    for j in range(L):
        b = np.stack([Gi[j] @ M[j][p] @ G[j + 1] for p in range(2)])
        nb = np.sqrt(np.einsum("pab,pab->", b, b.conj()).real / b.shape[1])
        B.append(b / nb)
    return B, S, eta
