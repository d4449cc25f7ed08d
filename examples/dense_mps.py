"""Small NumPy helpers for the example scripts: they check the converted MPS from its dense site tensors (what the
reference's examples do with TeNPy's ``MPS.correlation_function`` / ``MPS.overlap``).  Not part of the package."""
import numpy as np


def state_tensors(mps):
    """Site tensors (2, chi_l, chi_r) whose product is the state: the Schmidt values of the orthogonality centre (the
    bond between the last 'A' and the first 'B' tensor) multiplied into the tensor to its right (or left, at the end)."""
    T = [np.asarray(t) for t in mps.dense_tensors()]
    form = list(mps.form)
    oc = sum(f == "A" for f in form)
    if 0 < oc < len(T):
        T[oc] = T[oc] * np.asarray(mps.lam[oc])[None, :, None]
    return T


def _step(E, a, b, op=None):
    """E'[c, d] = sum_pq op[p, q] conj(a[p])^T E b[q] (op = identity if None)."""
    out = 0
    for p in range(2):
        for q in range(2):
            w = (1.0 if p == q else 0.0) if op is None else op[p, q]
            if w != 0:
                out = out + w * (np.conj(a[p]).T @ E @ b[q])
    return out


def overlap(Ta, Tb):
    """<a|b> of two finite MPS given by their state tensors."""
    E = np.ones((1, 1), complex)
    for a, b in zip(Ta, Tb):
        E = _step(E, a, b)
    return E[0, 0]


def correlation_function(T, kind="CdC"):
    """<c^dagger_i c_j> (kind 'CdC') or <c_i c_j> (kind 'CC') of a fermionic MPS with Jordan-Wigner strings, O(L^2 chi^3)."""
    L = len(T)
    n = np.array([[0, 0], [0, 1.0]])
    cd = np.array([[0, 0], [1.0, 0]])      # |1><0|
    c = cd.T
    Z = np.diag([1.0, -1.0])
    first = (cd if kind == "CdC" else c) @ Z
    R = [None] * (L + 1)                    # right environments: R[i][a, b] = sum_p conj(T[p])[a, c] R[c, d] T[p][b, d]
    R[L] = np.ones((1, 1), complex)
    for i in range(L - 1, -1, -1):
        R[i] = sum(np.conj(T[i][p]) @ R[i + 1] @ T[i][p].T for p in range(2))
    out = np.zeros((L, L), complex)
    E = np.ones((1, 1), complex)
    for i in range(L):
        if kind == "CdC":
            out[i, i] = np.sum(_step(E, T[i], T[i], n) * R[i + 1])
        F = _step(E, T[i], T[i], first)
        for j in range(i + 1, L):
            out[i, j] = np.sum(_step(F, T[j], T[j], c) * R[j + 1])
            F = _step(F, T[j], T[j], Z)
        E = _step(E, T[i], T[i])
    if kind == "CdC":
        out = out + np.triu(out, 1).conj().T
    else:
        out = out - np.triu(out, 1).T
    return out
