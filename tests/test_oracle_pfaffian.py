"""Pins oracle/pfaffian_oracle.py against fixtures produced by the reference's own NumPy core
(tests/golden/make_golden_pfaffian.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import pfaffian_oracle as porc

NAMES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("pf_") and f.endswith(".npz"))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def test_pfaffian_standin_properties():
    """The Parlett-Reid routine that stands in for pfapack: Pf^2 = det, Pf(B A B^T) = det(B) Pf(A),
    closed forms for 2x2 / 4x4."""
    rng = np.random.default_rng(0)
    assert porc.pfaffian(np.array([[0, 3.0], [-3.0, 0]])) == 3.0
    a, b, c, d, e, f = rng.standard_normal(6)
    A4 = np.array([[0, a, b, c], [-a, 0, d, e], [-b, -d, 0, f], [-c, -e, -f, 0]])
    assert abs(porc.pfaffian(A4) - (a * f - b * e + c * d)) < 1e-14
    for n in (2, 6, 10, 16):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        A = M - M.T
        pf = porc.pfaffian(A)
        assert abs(pf**2 - np.linalg.det(A)) < 1e-9 * abs(np.linalg.det(A))
        B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        assert abs(porc.pfaffian(B @ A @ B.T) - np.linalg.det(B) * pf) < 1e-9 * abs(np.linalg.det(B) * pf)
    assert porc.pfaffian(np.zeros((3, 3))) == 0.0


@pytest.mark.parametrize("name", NAMES)
def test_pfaffian_oracle_matches_reference_fixture(name):
    g = load(name)
    kw = {}
    if "kw_ortho_center" in g:
        kw["ortho_center"] = int(g["kw_ortho_center"])
    C = porc.correlation_matrix(g["H"])
    np.testing.assert_allclose(C, g["C"], rtol=0, atol=1e-13)
    cuts, sites = porc.c_to_mps(g["C"], {"chi_max": int(g["chi_max"])}, **kw)
    L, oc = int(g["L"]), int(g["ortho_center"])
    assert cuts[oc].parity() == int(g["parity"])
    for b in range(L + 1):
        c = cuts[b]
        np.testing.assert_allclose(c.e, g[f"b{b}_e"], rtol=0, atol=1e-13)
        np.testing.assert_array_equal([-1 if c.pL is None else c.pL, -1 if c.pR is None else c.pR], g[f"b{b}_p"])
        np.testing.assert_array_equal(c.sets, g[f"b{b}_sets"])
        np.testing.assert_allclose(c.lam_raw, g[f"b{b}_lam_raw"], rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(sorted(c.idx_n), g[f"b{b}_n"])
        np.testing.assert_array_equal([c.idx_n[k][0] for k in sorted(c.idx_n)], g[f"b{b}_nstart"])
    for i in range(L):
        s = sites[i]
        assert s.qtotal == int(g[f"s{i}_qtotal"])
        np.testing.assert_array_equal(s.sets_bra, g[f"s{i}_sets_bra"])
        np.testing.assert_array_equal(s.sets_ket, g[f"s{i}_sets_ket"])
        np.testing.assert_array_equal(s.leg_idx_bra, g[f"s{i}_leg_idx_bra"])
        np.testing.assert_allclose(s.norm, g[f"s{i}_norm"], rtol=1e-10)
        np.testing.assert_allclose(s.N, g[f"s{i}_N"], rtol=0, atol=1e-10)
        keys = g[f"s{i}_blkkeys"]
        assert len(keys) == len(s.blocks)
        for nb, r0, r1, nk, c0, c1 in keys:
            b0, b1, k0, k1, blk = s.blocks[(int(nb), int(nk))]
            assert (b0, b1, k0, k1) == (int(r0), int(r1), int(c0), int(c1))
            ref = g[f"s{i}_blk_{nb}_{nk}"]
            np.testing.assert_allclose(blk, ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", NAMES)
def test_pfaffian_oracle_canonical_form(name):
    """Right-/left-canonical isometry of the assembled tensors where the ket bond is untruncated
    (the DEBUG check of pfaffian.py:1876-1881, 1907-1912), via the merged-leg positions."""
    g = load(name)
    kw = {}
    if "kw_ortho_center" in g:
        kw["ortho_center"] = int(g["kw_ortho_center"])
    cuts, sites = porc.c_to_mps(g["C"], {"chi_max": int(g["chi_max"])}, **kw)
    L = int(g["L"])
    checked = 0
    for i, s in enumerate(sites):
        bra, ket = (cuts[i], cuts[i + 1]) if s.mode == "left" else (cuts[i + 1], cuts[i])
        chi_b, chi_k = len(bra.lam), len(ket.lam)
        if chi_k < 2 ** ket.k or chi_b < 2 ** bra.k:
            continue  # truncated bond: not an isometry
        T = np.zeros((2 * chi_b, chi_k), complex)
        for (r0, r1, c0, c1, blk) in s.blocks.values():
            T[s.leg_idx_bra[r0:r1], c0:c1] = blk
        G = T.conj().T @ T
        np.testing.assert_allclose(G, np.eye(chi_k), atol=1e-8)
        checked += 1
    assert checked >= 2


def test_pfaffian_oracle_cut_decomposition_at_mid_size():
    """Every bond of a random BdG chain of 64 sites (chi_max = 64, where chi_max truncates most bonds) against the
    reference's own cut decomposition (tests/golden/make_golden_summary.py pf_randbdg_L64_s5_chi64: pfaffian.py:685-920 and
    :1008-1248 run unmodified, no sub-Pfaffian involved): chi, eigenvalues, vacuum parities, patterns, Schmidt values, S."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden_pfaffian import random_majorana_H
    ref = np.load(os.path.join(GOLDEN, "full", "pf_randbdg_L64_s5_chi64.npz"))
    L, oc = int(ref["L"]), int(ref["ortho_center"])
    C = porc.correlation_matrix(random_majorana_H(L, 5))
    trunc = porc.as_trunc({"chi_max": int(ref["chi_max"])})
    centre = porc.cut_vectors(C, oc, trunc, "LR")
    parity = centre.parity()
    assert parity == int(ref["total_parity"])
    for b in range(L + 1):
        c = centre if b == oc else porc.cut_vectors(C, b, trunc, "L" if b < oc else "R", parity)
        e_ref = ref["e"][ref["e_off"][b]: ref["e_off"][b + 1]]
        lam_ref = ref["lam"][ref["lam_off"][b]: ref["lam_off"][b + 1]]
        assert len(c.lam) == int(ref["chi"][b])
        np.testing.assert_allclose(c.e, e_ref, rtol=0, atol=1e-13)
        pL, pR = (int(x) for x in ref["parities"][b])
        assert (-1 if c.pL is None else c.pL) == pL and (-1 if c.pR is None else c.pR) == pR
        packed = ref["sets_packed"][ref["sets_off"][b]: ref["sets_off"][b + 1]].reshape(len(lam_ref), -1)
        sets_ref = np.unpackbits(packed, axis=1, bitorder="little")[:, : len(e_ref)].astype(bool)
        np.testing.assert_array_equal(c.sets, sets_ref)
        np.testing.assert_allclose(c.lam, lam_ref, rtol=0, atol=1e-12)
        p = c.lam ** 2
        assert abs(-(p[p > 0] * np.log(p[p > 0])).sum() - float(ref["S"][b])) < 1e-12
        idx = ref["idx_n"][ref["idx_n_off"][b]: ref["idx_n_off"][b + 1]].reshape(-1, 3)
        assert sorted(c.idx_n) == [int(x) for x in idx[:, 0]]
        assert [list(c.idx_n[int(k)]) for k in idx[:, 0]] == [[int(a), int(z)] for _, a, z in idx]
