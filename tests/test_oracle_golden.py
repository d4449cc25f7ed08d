"""Pins oracle/slater_oracle.py against fixtures produced by the reference's own
NumPy core (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from oracle import slater_oracle as orc


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def run_oracle(g):
    kw = {}
    if "kw_ortho_center" in g:
        kw["ortho_center"] = int(g["kw_ortho_center"])
    if "kw_spinful" in g:
        kw["spinful"] = str(g["kw_spinful"])
    return orc.c_to_mps(g["C_in"], {"chi_max": int(g["chi_max"])}, **kw)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_fixture(name):
    g = load(name)
    C_in, N = orc.correlation_matrix(g["H"])
    np.testing.assert_allclose(C_in, g["C_in"], rtol=0, atol=1e-14)
    assert N == int(g["N"])
    cuts, sites = run_oracle(g)
    L, oc = int(g["L"]), int(g["ortho_center"])
    assert len(sites) == L
    for b in range(L + 1):
        c = cuts[b]
        # integers: exact
        np.testing.assert_array_equal(c.sets, g[f"b{b}_sets"])
        np.testing.assert_array_equal(sorted(c.sectors), g[f"b{b}_q"])
        np.testing.assert_array_equal([c.sectors[q][0] for q in sorted(c.sectors)], g[f"b{b}_qstart"])
        np.testing.assert_array_equal([c.sectors[q][1] for q in sorted(c.sectors)], g[f"b{b}_qstop"])
        np.testing.assert_array_equal([c.n_filled("L"), c.n_filled("R")], g[f"b{b}_nfilled"])
        # floats
        np.testing.assert_allclose(c.e, g[f"b{b}_e"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(c.lam_raw, g[f"b{b}_lam_raw"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(c.lam, g[f"b{b}_lam"], rtol=1e-12, atol=1e-15)
    for i in range(L):
        s = sites[i]
        np.testing.assert_array_equal(s.sets_bra, g[f"s{i}_sets_bra"])
        np.testing.assert_array_equal(s.sets_ket, g[f"s{i}_sets_ket"])
        assert s.qtotal == int(g[f"s{i}_qtotal"])
        np.testing.assert_allclose(s.det_always, g[f"s{i}_det_always"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(s.M, g[f"s{i}_M"], rtol=0, atol=1e-11)
        np.testing.assert_array_equal(sorted(s.blocks), sorted(g[f"s{i}_blkq"]))
        for q, r0, r1 in zip(g[f"s{i}_blkq"], g[f"s{i}_blkrow0"], g[f"s{i}_blkrow1"]):
            b0, b1, c0, c1, blk = s.blocks[int(q)]
            assert (b0, b1) == (int(r0), int(r1))
            ref = g[f"s{i}_blk{q}"]
            np.testing.assert_allclose(blk, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name,tol", [("rand_L16_s1_chi200", 1e-9),  # untruncated up to svd_min
                                      ("chain_L16_chi32", 1e-5), ("chainPH_L8_chi64", 1e-4),
                                      ("rand_L24_s3_oc7_chi48", 2e-3), ("chain_L8_chi8", 2e-3)])
def test_oracle_mps_reproduces_correlation_matrix(name, tol):
    """Acceptance check of src/examples/slater.py:30-36 on the oracle's tensors."""
    g = load(name)
    cuts, sites = run_oracle(g)
    oc = int(g["ortho_center"])
    T = orc.dense_tensors(cuts, sites)
    G = orc.mps_correlation(T, cuts[oc].lam, oc)
    # the tolerance is the chi_max truncation error of the case, nothing else
    np.testing.assert_allclose(G, g["C"], rtol=0, atol=tol)
    nrm = orc.mps_overlap(T, cuts[oc].lam, T, cuts[oc].lam, oc)
    assert abs(nrm - 1) < tol


def test_entropy_matches_free_fermion_formula():
    """S(b) = -sum e ln e + (1-e) ln(1-e) when untruncated."""
    g = load("chain_L32_chi200")
    cuts, _ = run_oracle(g)
    S = orc.entropies(cuts)
    for b in (4, 16, 28):
        e = cuts[b].e
        exact = -(e * np.log(e) + (1 - e) * np.log(1 - e)).sum()
        assert abs(S[b] - exact) < 2e-5  # chi truncation at svd_min
    assert abs(S[16] - 0.846819) < 1e-6  # BASELINE.md, config 1 centre bond


def test_lowest_sums_edge_cases():
    t = orc.as_trunc({"chi_max": 4})
    s, sets, _ = orc.lowest_sums([], t)
    assert sets.shape == (1, 0)
    s, sets, _ = orc.lowest_sums([], orc.as_trunc({"chi_max": 4, "sectors": 3}), filled_left=2)
    assert sets.shape == (0, 0)
    s, sets, _ = orc.lowest_sums([0.5, -1.0, 2.0], orc.as_trunc({"chi_max": 100}))
    assert len(s) == 8 and np.all(np.diff(s) >= 0)
    np.testing.assert_array_equal(sets[0], [False, True, False])
    # a degenerate pair straddling chi_max is dropped whole (schmidt_utils.py:163-185)
    s, sets, _ = orc.lowest_sums([1.0, 1.0, 3.0], orc.as_trunc({"chi_max": 2}))
    assert len(s) == 1
