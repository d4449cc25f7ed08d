"""CPU tests of the Gutzwiller oracle (oracle/gutzwiller_oracle.py): the restated group / project /
canonical_form_finite pipeline against the independent brute-force check (projector applied to the full
state vector, one dense SVD per cut).  TeNPy itself is not installed anywhere we can run, so its arithmetic
is 'parity unpinned'; everything gauge invariant is pinned here."""
import numpy as np
import pytest

from oracle import gutzwiller_oracle as gw
from oracle import slater_oracle as orc
from tests_inputs import random_hopping, uniform_chain


def fermion_mps(H, chi, spinful, oc=None):
    C, _ = orc.correlation_matrix(H)
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi}, ortho_center=oc, spinful=spinful)
    T = orc.dense_tensors(cuts, sites)
    q = [np.concatenate([np.full(b - a, n) for n, (a, b) in sorted(c.sectors.items())]).astype(int) for c in cuts]
    oc = oc or len(T) // 2
    return T, q, cuts[oc].lam, oc


CASES = [("ph", uniform_chain(6) + 0.3 * np.diag(np.arange(6) % 3 - 1.0), "PH", None),
         ("ph", random_hopping(6, 3).real, "PH", 3),
         ("ph", random_hopping(6, 1), "PH", None),
         ("std", random_hopping(6, 2), "simple", None),
         ("std", uniform_chain(6) + 0.2 * np.diag(np.arange(6) - 2.5), "simple", 5)]


@pytest.mark.parametrize("kind,H,spinful,oc", CASES)
def test_restatement_against_brute_force(kind, H, spinful, oc):
    T, q, lam, oc = fermion_mps(H, 4096, spinful, oc)
    psi = gw.state_vector(T, lam, oc)
    assert abs(np.linalg.norm(psi) - 1) < 1e-10
    chi = gw.project_state(psi, kind)
    S_ref = gw.schmidt_values_of_state(chi)
    M, keep = gw.group_and_project(T, q, lam, oc, kind)
    B, S, nrm = gw.canonical_form_finite(M)
    assert abs(nrm - np.linalg.norm(chi)) < 1e-12
    for b, (a, r) in enumerate(zip(S, S_ref)):
        a = np.sort(a)[::-1]
        n = min(len(a), len(r))
        assert abs(len(a) - len(r)) <= 2 and np.abs(a[:n] - r[:n]).max() < 1e-12, b
    ov = abs(np.vdot(gw.state_vector(B), chi / np.linalg.norm(chi)))
    assert abs(ov - 1) < 1e-12
    for t in B:   # right-canonical
        X = np.einsum("pab,pcb->ac", t, t.conj())
        assert np.abs(X - np.eye(len(X))).max() < 1e-12
    if kind == "ph":   # charge-block version: same Schmidt values, definite 2 S^z per index
        ql = gw.spin_charges(q, keep)
        B2, S2, Q2, n2 = gw.canonical_form_finite_blocks(M, ql)
        assert abs(n2 - nrm) < 1e-12
        for b, (a, r) in enumerate(zip(S2, S_ref)):
            a = np.sort(a)[::-1]
            n = min(len(a), len(r))
            assert np.abs(a[:n] - r[:n]).max() < 1e-12, b
        assert Q2[0].tolist() == [0] and Q2[-1].tolist() == [0]
        assert abs(abs(np.vdot(gw.state_vector(B2), chi / np.linalg.norm(chi))) - 1) < 1e-12


def test_heisenberg_like_weight():
    """Half-filled uniform chain, PH-projected: the projected weight is the probability of no singly occupied
    site; it must lie strictly between 0 and 1 and the spin state must have total S^z = 0."""
    T, q, lam, oc = fermion_mps(uniform_chain(6), 4096, "PH")
    chi = gw.project_state(gw.state_vector(T, lam, oc), "ph")
    w = np.linalg.norm(chi) ** 2
    assert 0.0 < w < 1.0
    idx = np.argwhere(np.abs(chi) > 1e-12)
    assert np.all(idx.sum(axis=1) == 3)


def test_infer_parities_host_logic():
    """temfpy_amd.gutzwiller.infer_parities (host logic, no GPU): recovers the charge parities of a
    number-conserving MPS from its tensors and rejects tensors that mix parities."""
    from temfpy_amd.gutzwiller import infer_parities

    T, q, lam, oc = fermion_mps(random_hopping(6, 1), 64, "PH")
    got = infer_parities(T)
    assert all(np.array_equal(a, np.asarray(b) % 2) for a, b in zip(got, q))
    bad = [t.copy() for t in T]
    bad[3][1] += bad[3][0]
    with pytest.raises(ValueError, match="parity"):
        infer_parities(bad)


def test_infinite_canonical_form_against_a_long_finite_chain():
    """oracle.canonical_form_infinite pinned independently of its own arithmetic: the Schmidt values in the middle of a
    long finite chain of repeated cells (oracle.canonical_form_finite: QR + SVD sweeps, no Gram matrices) are those of the
    infinite MPS once the chain is much longer than the correlation length; the result is right-canonical, its left
    environment is diag(S^2) and it is the same state per unit cell."""
    rng = np.random.default_rng(5)
    chi, L, N = 5, 3, 80
    M = [rng.normal(size=(2, chi, chi)) + 1j * rng.normal(size=(2, chi, chi)) for _ in range(L)]
    B, S, eta = gw.canonical_form_infinite(M)
    M = [m / eta ** (0.5 / L) for m in M]
    for j, t in enumerate(B):
        X = np.einsum("pab,pcb->ac", t, t.conj())
        assert np.abs(X - np.eye(len(X))).max() < 1e-12
        l2 = sum(t[p].conj().T @ np.diag(S[j] ** 2) @ t[p] for p in range(2))
        assert np.abs(l2 - np.diag(S[j + 1] ** 2)).max() < 1e-12
    chain = [m for _ in range(N) for m in M]
    chain[0] = np.einsum("a,pab->pb", rng.normal(size=chi), chain[0])[:, None, :]
    chain[-1] = np.einsum("pab,b->pa", chain[-1], rng.normal(size=chi))[:, :, None]
    _, Sf, _ = gw.canonical_form_finite(chain, cutoff=1e-14)
    mid = L * (N // 2)
    for j in range(L):
        s = np.sort(Sf[mid + j])[::-1]
        assert np.abs(s[: len(S[j])] - np.sort(S[j])[::-1]).max() < 1e-10

    def mixed(A, C):
        E = None
        for a, c in zip(A, C):
            e = np.einsum("pab,pcd->acbd", a, np.conj(c)).reshape(a.shape[1] * c.shape[1], a.shape[2] * c.shape[2])
            E = e if E is None else E @ e
        return np.abs(np.linalg.eigvals(E)).max()
    assert abs(mixed(B, M) / np.sqrt(mixed(B, B) * mixed(M, M)) - 1) < 1e-10


def test_cell_projection_masks():
    """group_and_project_cell: the closing bond keeps the indices bond 0 keeps (idx_next = (idx + 1) % L, gutzwiller.py:235)."""
    rng = np.random.default_rng(1)
    q = [np.array([0, 0, 1, 1, 2]), np.array([0, 1, 1, 2, 2, 3]), np.array([1, 1, 2, 2, 3]), np.array([1, 2, 2, 3, 3, 4])]
    T = [rng.normal(size=(2, len(q[i]), len(q[(i + 1) % 4]))) for i in range(4)]
    M, keep = gw.group_and_project_cell(T, q, 2, "std", "N", 0, 1)
    assert keep[0].tolist() == [2, 3] and keep[1].tolist() == [2, 3] and keep[2].tolist() == [2, 3]
    assert [m.shape for m in M] == [(2, 2, 2), (2, 2, 2)]
    np.testing.assert_allclose(M[0][1], (T[0][1] @ T[1][0])[np.ix_(keep[0], keep[1])])
    M, keep = gw.group_and_project_cell(T, q, 2, "ph", "N", 1, 0)
    assert keep[0].tolist() == [2, 3] and keep[1].tolist() == [0, 1, 4]
    np.testing.assert_allclose(M[1][0], (T[2][0] @ T[3][0])[np.ix_(keep[1], keep[2])])
