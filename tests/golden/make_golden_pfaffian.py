#!/usr/bin/env python3
"""Golden fixtures for the Pfaffian (BCS / Nambu) path, from the REFERENCE's own NumPy core.

Same loading recipe as make_golden.py (placeholder objects for the TeNPy names).  Two more
stand-ins are unavoidable and are stated in DESIGN.md as 'parity unpinned':
  * `pfapack.ctypes.pfaffian` (third-party, not installed, version unpinned in pyproject.toml:37;
    call site pfaffian.py:49,1425) is replaced by a NumPy Parlett-Reid Pfaffian written here;
  * `pfaffian._many_pfaffian` (pfaffian.py:1413-1426) uses `warnings.catch_warnings(category=...)`,
    which needs Python >= 3.11 and raises TypeError on this container's 3.10; it is replaced by the
    same loop without the warnings filter.
Everything else (Nambu diagonalisation, 1/2-mode handling, vacuum parities, `_pfaffian_matrix`,
index gathers) is executed by the reference's unmodified code.
"""
import os
import sys
import types
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pfaffian_parlett_reid(A, **_):
    """Pfaffian of a skew-symmetric matrix by skew Gaussian elimination with pivoting."""
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


def load_reference_pfaffian():
    slater, testing = mg.load_reference()
    pf_mod = types.ModuleType("pfapack.ctypes")
    pf_mod.pfaffian = pfaffian_parlett_reid
    pk = types.ModuleType("pfapack")
    pk.ctypes = pf_mod
    sys.modules["pfapack"], sys.modules["pfapack.ctypes"] = pk, pf_mod
    import importlib

    pfaffian = importlib.import_module("temfpy.pfaffian")

    def many(matrices, **kw):  # pfaffian.py:1413-1426 without the py3.11-only warnings filter
        shape = matrices.shape[:-2]
        m = matrices.reshape(int(np.prod(shape)), *matrices.shape[-2:])
        return np.asarray([pfaffian_parlett_reid(a) for a in m]).reshape(shape)

    pfaffian._many_pfaffian = many
    return pfaffian, testing


def random_majorana_H(L, seed, rng_range=3.0):
    """src/examples/pfaffian.py:13-17 with a seeded generator (Majorana basis, H = i(M - M^T))."""
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(2 * L), np.arange(2 * L), indexing="ij")
    M = rng.normal(size=(2 * L, 2 * L)) * np.exp(-abs(x - y) / rng_range)
    return 1j * (M - M.T)


def kitaev_majorana_H(L, t1, t2):
    """src/examples/iMPS_pfaffian.py:7-11: Majorana chain with alternating bonds t1, t2."""
    t = np.where(np.arange(2 * L - 1) % 2 == 0, t1, t2)
    M = np.diag(t, 1)
    return M + M.T.conj()


def half_mode_majorana_H(L, lefts, seed):
    """Majorana Hamiltonian whose entanglement cuts carry exact eigenvalue-1/2 modes (pfaffian.py:807-874):
    every Majorana j in `lefts` is dimerised with its mirror image 2L-1-j across the chain (the
    situation of a topological chain whose end modes are paired) and decoupled from a random BdG
    system on all the other Majoranas.  Every cut that separates a dimer sees its left member as an
    isolated 1/2 mode, and the remaining odd number of Majoranas forces a second one."""
    M = np.zeros((2 * L, 2 * L))
    out = sorted(set(lefts) | {2 * L - 1 - j for j in lefts})
    inner = np.array([j for j in range(2 * L) if j not in out])
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(inner, inner, indexing="ij")
    M[np.ix_(inner, inner)] = rng.normal(size=(len(inner), len(inner))) * np.exp(-abs(x - y) / 3.0)
    for i, j in enumerate(lefts):
        M[j, 2 * L - 1 - j] = 1.0 + 0.5 * i
    return 1j * (M - M.T)


CASES = [
    ("pf_rand_L6_s0_chi16", lambda: random_majorana_H(6, 0), dict(chi_max=16)),
    ("pf_rand_L8_s1_chi32", lambda: random_majorana_H(8, 1), dict(chi_max=32)),
    ("pf_rand_L10_s2_chi24", lambda: random_majorana_H(10, 2), dict(chi_max=24)),
    ("pf_rand_L9_s3_oc3_chi20", lambda: random_majorana_H(9, 3), dict(chi_max=20, ortho_center=3)),
    ("pf_kitaev_L8_trivial_chi16", lambda: kitaev_majorana_H(8, 1.5j, 1j), dict(chi_max=16)),
    ("pf_half_L8_p1_s0_chi32", lambda: half_mode_majorana_H(8, [0], 0), dict(chi_max=32)),
    ("pf_half_L9_p1_s1_oc3_chi24", lambda: half_mode_majorana_H(9, [1], 1), dict(chi_max=24, ortho_center=3)),
]


def replay(pf, C, chi_max, ortho_center=None):
    """pfaffian.C_to_MPS (pfaffian.py:1832-1914) with dense collection instead of TeNPy arrays."""
    SV, TD = pf.SchmidtVectors, pf.MPSTensorData
    trunc = {"chi_max": chi_max}
    L = len(C) // 2
    oc = ortho_center or L // 2
    out = {"C": C, "L": L, "ortho_center": oc, "chi_max": chi_max}

    def put_bond(b, S):
        m = S.modes
        out[f"b{b}_e"] = m.e
        out[f"b{b}_p"] = np.array([-1 if m.pL is None else m.pL, -1 if m.pR is None else m.pR])
        sets = S.left_sets if S.left_sets is not None else S.right_sets[:, ::-1]
        out[f"b{b}_sets"] = sets
        out[f"b{b}_lam_raw"] = S.schmidt_values
        out[f"b{b}_lam"] = S.schmidt_values / np.linalg.norm(S.schmidt_values)
        ks = np.array(sorted(S.idx_n))
        out[f"b{b}_n"] = ks
        out[f"b{b}_nstart"] = np.array([S.idx_n[k].start for k in ks])
        out[f"b{b}_nstop"] = np.array([S.idx_n[k].stop for k in ks])

    def put_site(i, T):
        out[f"s{i}_norm"] = np.asarray(T.norm)
        out[f"s{i}_N"] = T.pfaffian_matrix
        out[f"s{i}_sets_bra"] = T.new_sets_bra
        out[f"s{i}_sets_ket"] = T.new_sets_ket
        out[f"s{i}_leg_idx_bra"] = np.asarray(T.leg_idx_bra)
        out[f"s{i}_qtotal"] = np.array(T.qtotal)
        keys = []
        for n_bra, sl_b in T.idx_n_bra.items():  # pfaffian.py:1766-1776
            for n_ket, sl_k in T.idx_n_ket.items():
                if (n_bra + n_ket) % 2 == 1:
                    continue
                blk = T.norm * pf._tensor_block(T.pfaffian_matrix, T.new_sets_bra[sl_b], T.new_sets_ket[sl_k])
                out[f"s{i}_blk_{n_bra}_{n_ket}"] = blk
                keys.append((n_bra, sl_b.start, sl_b.stop, n_ket, sl_k.start, sl_k.stop))
        out[f"s{i}_blkkeys"] = np.array(keys).reshape(-1, 6)

    Sc = SV.from_correlation_matrix(C, oc, trunc, basis="M")
    put_bond(oc, Sc)
    parity = Sc.parity()
    out["parity"] = np.array(parity)
    S = Sc
    for i in range(oc, L):
        Sn = SV.from_correlation_matrix(C, i + 1, trunc, which="R", basis="M", total_parity=parity)
        put_bond(i + 1, Sn)
        put_site(i, TD.from_schmidt_vectors(Sn, S, "right"))
        S = Sn
    S = Sc
    for i in reversed(range(oc)):
        Sn = SV.from_correlation_matrix(C, i, trunc, which="L", basis="M", total_parity=parity)
        put_bond(i, Sn)
        put_site(i, TD.from_schmidt_vectors(Sn, S, "left"))
        S = Sn
    return out


def main():
    pf, testing = load_reference_pfaffian()
    warnings.simplefilter("ignore")
    only = sys.argv[1] if len(sys.argv) > 1 else ""   # optional name prefix: regenerate a subset
    for name, builder, kw in CASES:
        if not name.startswith(only):
            continue
        H = builder()
        C = pf.correlation_matrix(H, "M->M")
        data = replay(pf, C, **kw)
        data["H"] = H
        if "ortho_center" in kw:
            data["kw_ortho_center"] = np.array(kw["ortho_center"])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        oc = data["ortho_center"]
        print(f"{name}: L={data['L']} parity={data['parity']} chi@centre={len(data['b%d_lam' % oc])} "
              f"k@centre={len(data['b%d_e' % oc])} {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
