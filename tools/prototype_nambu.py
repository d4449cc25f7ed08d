"""Throw-away NumPy prototype: Nambu (BCS) cut solver from GEMM / QR / small eigenproblems only.

Replaces the full eigh of pfaffian.py:789 by
  * entangled + 1/2 modes: left singular vectors of the off-diagonal block with
    sigma^2 = lambda (1 - lambda) >= cutoff (1 - cutoff), Rayleigh-Ritz on the block;
  * "empty" (lambda ~ 0) basis: dominant subspace of (1 - A) orthogonal to those;
  * upper (lambda > 1/2) halves are the complex conjugates of the lower ones (Majorana basis).
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import pfaffian_oracle as porc
from oracle import slater_oracle as orc


def diag_nambu_subspace(c, off, cutoff, deg_tol, rng, want):
    """Same return convention as porc._diag_nambu: e (2n,) ascending, v (2n, 2n), ke, kh.
    Only the half that `want` ("L": lower, "R": upper) needs and the 1/2 block are filled in."""
    n2 = len(c)
    n = n2 // 2
    if n == 0:
        return np.zeros(0), np.zeros((0, 0), complex), 0, 0
    m2 = off.shape[1]
    e = np.zeros(n2)
    e[n:] = 1.0
    v = np.zeros((n2, n2), complex)
    if m2 > 0:
        p = min(64, n2, m2)
        Y = off @ (rng.standard_normal((m2, p)) + 1j * rng.standard_normal((m2, p)))
        Q, _ = np.linalg.qr(Y)
        B = Q.conj().T @ off
        Z, sv, _ = np.linalg.svd(B, full_matrices=False)
        keep = sv**2 >= cutoff * (1 - cutoff)
        U0 = Q @ Z[:, keep]
        T = U0.conj().T @ c @ U0
        lam, X = np.linalg.eigh((T + T.conj().T) / 2)
        UE = U0 @ X                        # ascending lambda, symmetric around 1/2
        assert len(lam) % 2 == 0
        ke = len(lam) // 2
        kh = int(np.sum(np.abs(lam - 0.5) <= deg_tol)) // 2
    else:
        lam, UE, ke, kh = np.zeros(0), np.zeros((n2, 0), complex), 0, 0
    # entangled lower / upper blocks next to the middle of the spectrum
    e[n - ke: n + ke] = lam
    v[:, n - ke: n + ke] = UE
    if kh:  # real basis of the 1/2 eigenspace (pfaffian.py:807-816)
        w = np.column_stack((UE[:, ke - kh: ke + kh].real, UE[:, ke - kh: ke + kh].imag))
        w, _, _ = np.linalg.svd(w)
        v[:, n - kh: n + kh] = w[:, : 2 * kh]
    # empty basis (lambda ~ 0): dominant subspace of (1 - A), orthogonal to everything found so far
    ne = n - ke
    if ne > 0:
        P = np.eye(n2) - c
        Yf = P @ (P @ (rng.standard_normal((n2, ne)) + 1j * rng.standard_normal((n2, ne))))
        Wk = v[:, n - ke: n + ke]
        Yf -= Wk @ (Wk.conj().T @ Yf)
        Qe, _ = np.linalg.qr(Yf)
        Qe -= Wk @ (Wk.conj().T @ Qe)
        Qe, _ = np.linalg.qr(Qe)
        if want == "L":
            v[:, :ne] = Qe
        else:  # filled = conj(empty); ascending order does not matter inside the (pruned) filled block
            v[:, n + ke:] = Qe.conj()
    return e, v, ke, kh


def c_to_mps_subspace(C, trunc, ortho_center=None, seed=3):
    rng = np.random.default_rng(seed)
    saved = porc._diag_nambu
    state = {}

    def patched_cut_modes(C_, x, tr, which="LR", total_parity=None, diag_tol=1e-8):
        Cn = (C_ + C_.conj().T) / 2
        Cn = 0.5 * np.eye(len(Cn)) + 1j * Cn.imag
        calls = []

        def dn(c, cutoff, deg_tol, diag_tol_):
            side = "L" if len(calls) == 0 and "L" in which else "R"
            calls.append(side)
            off = Cn[: 2 * x, 2 * x:] if side == "L" else Cn[2 * x:, : 2 * x]
            return diag_nambu_subspace(c, off, cutoff, deg_tol, rng, side)

        porc._diag_nambu = dn
        try:
            return state["orig"](C_, x, tr, which, total_parity, diag_tol)
        finally:
            porc._diag_nambu = saved

    state["orig"] = porc.cut_modes
    porc.cut_modes = patched_cut_modes
    try:
        return porc.c_to_mps(C, trunc, ortho_center)
    finally:
        porc.cut_modes = state["orig"]


def dense(cuts, sites):
    T = []
    for i, s in enumerate(sites):
        bra, ket = (cuts[i], cuts[i + 1]) if s.mode == "left" else (cuts[i + 1], cuts[i])
        cb, ck = len(bra.lam), len(ket.lam)
        M = np.zeros((2 * cb, ck), complex)
        for (r0, r1, c0, c1, blk) in s.blocks.values():
            M[s.leg_idx_bra[r0:r1], c0:c1] = blk
        t = M.reshape(2, cb, ck)
        T.append(t if s.mode == "left" else t.transpose(0, 2, 1))
    return T


def compare(H, chi, label, oc=None):
    C = porc.correlation_matrix(H)
    cr, sr = porc.c_to_mps(C, {"chi_max": chi}, oc)
    cs, ss = c_to_mps_subspace(C, {"chi_max": chi}, oc)
    L = len(C) // 2
    o = oc or L // 2
    same = all(np.array_equal(a.sets, b.sets) and a.pL == b.pL and a.pR == b.pR for a, b in zip(cr, cs))
    de = max(np.abs(a.e - b.e).max() for a, b in zip(cr, cs) if a.e.size and a.e.shape == b.e.shape)
    Tr, Ts = dense(cr, sr), dense(cs, ss)
    ov = abs(orc.mps_overlap(Tr, cr[o].lam, Ts, cs[o].lam, o)) / np.sqrt(
        abs(orc.mps_overlap(Tr, cr[o].lam, Tr, cr[o].lam, o)) * abs(orc.mps_overlap(Ts, cs[o].lam, Ts, cs[o].lam, o)))
    dn = max(abs(abs(a.norm) - abs(b.norm)) for a, b in zip(sr, ss))
    print(f"{label}: sets/parities equal={same} max|de|={de:.2e} max|d norm|={dn:.2e} 1-overlap={1-ov:.2e} kh max={max(c.kh for c in cr)}")


if __name__ == "__main__":
    sys.path.insert(0, "tests/golden")
    from make_golden_pfaffian import random_majorana_H, kitaev_majorana_H
    compare(random_majorana_H(6, 0), 16, "rand L=6")
    compare(random_majorana_H(10, 2), 24, "rand L=10")
    compare(random_majorana_H(16, 4), 48, "rand L=16")
    compare(random_majorana_H(24, 5), 64, "rand L=24")
    compare(random_majorana_H(9, 3), 20, "rand L=9 oc=3", 3)
    compare(kitaev_majorana_H(8, 1.5j, 1j), 16, "kitaev trivial L=8")
    compare(kitaev_majorana_H(12, 1.5j, 1j), 32, "kitaev trivial L=12")
