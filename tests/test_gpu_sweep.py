"""GPU parity tests of the whole Slater -> MPS sweep (HIP path through the C ABI) against
the oracle and the reference-generated fixtures.

Tolerances (fp64): entangled eigenvalues 1e-13 abs, Schmidt values 1e-9 abs, entropies
1e-10, |tensor entries| 1e-7 * max (weak orbitals are only determined to ~1e-10 by C itself),
state overlap 1 - |<ref|hip>| <= 1e-9.  Integer outputs (chi, subsets, sector slices) exact
for non-degenerate spectra; for exactly degenerate spectra the order inside a degenerate
multiplet depends on rounding noise (also in the reference), so they are compared as sets
and through gauge-invariant quantities."""
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from oracle import slater_oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def run_hip(C, chi, **kw):
    from temfpy_amd import slater

    return slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False, **kw)


def overlap(cuts, sites, mps, oc):
    T1 = orc.dense_tensors(cuts, sites)
    n1 = orc.mps_overlap(T1, cuts[oc].lam, T1, cuts[oc].lam, oc)
    T2 = mps.dense_tensors()
    n2 = orc.mps_overlap(T2, mps.lam[oc], T2, mps.lam[oc], oc)
    return abs(orc.mps_overlap(T1, cuts[oc].lam, T2, mps.lam[oc], oc)) / np.sqrt(abs(n1 * n2))


NONDEGENERATE = [n for n in golden_names() if n.startswith("rand_")]
DEGENERATE = [n for n in golden_names() if not n.startswith("rand_")]


@pytest.mark.parametrize("name", NONDEGENERATE)
def test_sweep_matches_reference_fixture(name):
    g = load(name)
    kw = {}
    if "kw_ortho_center" in g:
        kw["ortho_center"] = int(g["kw_ortho_center"])
    mps = run_hip(g["C_in"], int(g["chi_max"]), **kw)
    L, oc = int(g["L"]), int(g["ortho_center"])
    assert mps.L == L and mps.ortho_center == oc
    for b in range(L + 1):
        bd = mps.bonds[b]
        np.testing.assert_array_equal(bd.sets, g[f"b{b}_sets"])
        np.testing.assert_array_equal([bd.n_filled_left, bd.n_filled_right], g[f"b{b}_nfilled"])
        qs, start = np.unique(bd.q_left, return_index=True)
        np.testing.assert_array_equal(qs, g[f"b{b}_q"])
        np.testing.assert_array_equal(start, g[f"b{b}_qstart"])
        np.testing.assert_allclose(bd.e, g[f"b{b}_e"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(bd.lam, g[f"b{b}_lam"], rtol=0, atol=1e-9)
    for i in range(L):
        s = mps.sites[i]
        assert sorted(b[0] for b in s.blocks) == sorted(int(q) for q in g[f"s{i}_blkq"])
        # det_always alone depends on the (arbitrary) basis of the filled subspace whenever the bra and
        # ket always-blocks differ in size; det_always * minor (= the blocks) is the invariant quantity
        for q, r0, r1, c0, c1, blk in s.blocks:
            ref = g[f"s{i}_blk{q}"]
            assert blk.shape == ref.shape
            np.testing.assert_allclose(np.abs(blk), np.abs(ref), rtol=0, atol=1e-7 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", NONDEGENERATE)
def test_sweep_state_overlap_with_oracle(name):
    g = load(name)
    kw = {}
    if "kw_ortho_center" in g:
        kw["ortho_center"] = int(g["kw_ortho_center"])
    chi = int(g["chi_max"])
    cuts, sites = orc.c_to_mps(g["C_in"], {"chi_max": chi}, **kw)
    mps = run_hip(g["C_in"], chi, **kw)
    oc = int(g["ortho_center"])
    assert abs(1 - overlap(cuts, sites, mps, oc)) < 1e-9
    dS = np.abs(orc.entropies(cuts) - mps.entanglement_entropy(all_bonds=True)).max()
    assert dS < 1e-10


@pytest.mark.parametrize("name", DEGENERATE)
def test_sweep_degenerate_spectra(name):
    """Uniform / SSH / spinful chains: exactly degenerate Schmidt multiplets."""
    g = load(name)
    kw = {}
    if "kw_spinful" in g:
        kw["spinful"] = str(g["kw_spinful"])
    chi = int(g["chi_max"])
    cuts, sites = orc.c_to_mps(g["C_in"], {"chi_max": chi}, **kw)
    mps = run_hip(g["C_in"], chi, **kw)
    L, oc = int(g["L"]), int(g["ortho_center"])
    same_basis = True
    for b in range(L + 1):
        np.testing.assert_allclose(mps.bonds[b].e, cuts[b].e, rtol=0, atol=1e-13)
        if mps.bonds[b].chi != len(cuts[b].lam):
            same_basis = False  # a degenerate multiplet straddles chi_max: rounding decides
            continue
        if sorted(map(bytes, mps.bonds[b].sets)) != sorted(map(bytes, cuts[b].sets)):
            same_basis = False  # chi_max cuts a pair that is degenerate only in exact arithmetic
        np.testing.assert_allclose(np.sort(mps.bonds[b].lam), np.sort(cuts[b].lam), rtol=0, atol=1e-8)
    if same_basis and name != "randSimple_L6_s4_chi32":
        assert abs(1 - overlap(cuts, sites, mps, oc)) < 1e-9
        dS = np.abs(orc.entropies(cuts) - mps.entanglement_entropy(all_bonds=True)).max()
        assert dS < 1e-9
    # physics check of src/examples/slater.py:30-36: same accuracy as the oracle's MPS
    G_ref = orc.mps_correlation(orc.dense_tensors(cuts, sites), cuts[oc].lam, oc)
    G_hip = orc.mps_correlation(mps.dense_tensors(), mps.lam[oc], oc)
    err_ref = np.abs(G_ref - g["C"]).max()
    err_hip = np.abs(G_hip - g["C"]).max()
    assert err_hip <= 2 * err_ref + 1e-9


def test_config1_uniform_chain_entropies():
    """BASELINE config 1: L=32 uniform chain, chi_max=200 -> chi<=71, S(centre)=0.846819."""
    g = load("chain_L32_chi200")
    mps = run_hip(g["C_in"], 200)
    S = mps.entanglement_entropy(all_bonds=True)
    assert max(mps.chi) == 71
    assert abs(S[16] - 0.846819) < 1e-6
    assert abs(S[1] - 0.693147) < 1e-6
    np.testing.assert_allclose(S, S[::-1], atol=1e-9)


@pytest.mark.parametrize("L,chi,seed", [(64, 64, 1), (96, 128, 0)])
def test_sweep_larger_random(L, chi, seed):
    from tests_inputs import random_hopping

    C, _ = orc.correlation_matrix(random_hopping(L, seed))
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi})
    mps = run_hip(C, chi)
    for b in range(L + 1):
        np.testing.assert_array_equal(mps.bonds[b].sets, cuts[b].sets)
        np.testing.assert_allclose(mps.bonds[b].lam, cuts[b].lam, rtol=0, atol=1e-9)
    assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9
    assert np.abs(orc.entropies(cuts) - mps.entanglement_entropy(all_bonds=True)).max() < 1e-10


def test_config2_properties_L256_chi128():
    """BASELINE config 2 at full size: size-independent properties (no oracle run)."""
    from tests_inputs import random_hopping

    L, chi = 256, 128
    C, N = orc.correlation_matrix(random_hopping(L, 0))
    mps = run_hip(C, chi)
    assert max(mps.chi) == chi and mps.chi[0] == 1 and mps.chi[-1] == 1
    S = mps.entanglement_entropy(all_bonds=True)
    assert abs(S[L // 2] - 0.818861000) < 1e-6  # BASELINE.md provisional golden
    for b in (1, 17, 100, 128, 200, 255):
        bd = mps.bonds[b]
        assert abs((bd.lam**2).sum() - 1) < 1e-12
        # charge bookkeeping: every kept vector has N_left + N_right = N
        assert np.all(bd.q_left == bd.n_filled_left + bd.sets.sum(axis=1))
    # right-canonical isometry on untruncated bonds near the right end
    T = mps.dense_tensors()
    for i in range(L - 4, L):
        B = T[i]
        G = np.einsum("pab,pcb->ac", B, B.conj())
        np.testing.assert_allclose(G, np.eye(len(G)), atol=1e-9)
    # tensor norm ~ 1 (slater.py:1313), sites in the bulk are chi-truncated
    for i in (10, 128, 250):
        nrm = mps.sites[i].norm() / np.sqrt(mps.chi[i] if i >= L // 2 else mps.chi[i + 1])
        assert 0.85 < nrm <= 1 + 1e-9


def test_argument_errors_match_reference():
    from temfpy_amd import slater

    C = np.eye(4) * 0.5
    with pytest.raises(ValueError):
        slater.C_to_MPS(C, {"chi_max": 4}, spinful="nope")
    with pytest.raises(ValueError):
        slater.C_to_MPS(C, {"chi_max": 4}, unit_cell_width=3)
    with pytest.raises(TypeError):
        slater.C_to_MPS(C, [1, 2])
    with pytest.raises(AssertionError):
        slater.C_to_MPS(C, {"chi_max": 0})


def test_site_sharding_reproduces_unsharded_result_bitwise():
    """Multi-GPU path (bench.py --gpus N): every rank converts a contiguous site range and
    recomputes the cut on its boundary.  The kernels are deterministic, so shards must
    reproduce the single-GPU tensors bit for bit (checked here on one GPU, shard by shard)."""
    from tests_inputs import random_hopping
    from temfpy_amd.engine import Engine
    from temfpy_amd.schmidt_utils import to_stopping_condition
    from temfpy_amd.multi_gpu import shard_sites

    L, chi = 48, 32
    C, _ = orc.correlation_matrix(random_hopping(L, 5))
    tr = to_stopping_condition({"chi_max": chi})
    eng = Engine("cuda:0")
    full = eng.run(C, tr, L // 2, L)
    for world in (2, 3, 4):
        ranges = shard_sites(L, L // 2, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == L
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        for (lo, hi) in ranges:
            part = eng.run(C, tr, L // 2, L, site_range=(lo, hi))
            for i in range(lo, hi):
                assert len(part.sites[i].blocks) == len(full.sites[i].blocks)
                for bp, bf in zip(part.sites[i].blocks, full.sites[i].blocks):
                    assert bp[:5] == bf[:5]
                    assert np.array_equal(bp[5], bf[5])
            for b in range(lo, hi + 1):
                assert np.array_equal(part.bonds[b].lam, full.bonds[b].lam)
                assert np.array_equal(part.bonds[b].masks, full.bonds[b].masks)


def test_site_sharding_bitwise_when_shards_take_other_kernel_variants():
    """tests/soak/soak_shards.py seed 60313 (round 3): a real spinful chain of 122 sites, chi_max 300, cut into the ranges
    (0, 1), (1, 7), (7, 122).  The short ranges next to the chain's end have only small charge sectors, so their launches
    choose the 32-bit-mask determinant kernel (and, until this was separated, the register form of the slab QR) where the
    unsharded conversion uses the 64-bit one: the variants must do the same arithmetic pair by pair - they differed by 1 ulp in
    pairs of order 5 (closed form in one kernel, queued elimination in the other)."""
    from temfpy_amd import slater
    from temfpy_amd.engine import Engine
    from temfpy_amd.schmidt_utils import to_stopping_condition

    rng = np.random.default_rng(60313)
    L = int(rng.integers(4, 65)); rng_h = float(rng.choice([0.7, 1.5, 3.0, 6.0])); cplx = bool(rng.integers(0, 2))
    x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
    H = M[0] + (1j * M[1] if cplx else 0)
    H = H + H.conj().T
    assert (L, cplx) == (61, False)
    C, _ = slater.correlation_matrix(H)
    C = slater.spinful_correlation_matrix(C, True)
    Lf, oc = len(C), 61
    tr = to_stopping_condition({"chi_max": 300})
    eng = Engine("cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = eng.run(C, tr, oc, Lf)
        for (lo, hi) in [(0, 1), (1, 7), (7, 40)]:
            part = eng.run(C, tr, oc, Lf, site_range=(lo, hi))
            for i in range(lo, hi):
                for bp, bf in zip(part.sites[i].blocks, full.sites[i].blocks):
                    assert bp[:5] == bf[:5] and np.array_equal(bp[5], bf[5]), (lo, hi, i, bp[:5])


def test_site_sharding_bitwise_on_the_wide_range_finder():
    """A spinful chain whose central cuts carry more than 64 entangled orbitals (range finder at 128 columns, block Jacobi
    in global memory), cut into two ranges: bit-identical to the unsharded conversion.  The block width of that Jacobi
    kernel used to follow the LARGEST problem of the launch, so a shard and the whole chain rotated in different orders
    (1e-16 apart, different bases inside the exactly degenerate multiplets of the two spin species:
    tests/soak/soak_shards.py seed 1952, whose input this is)."""
    import warnings

    from temfpy_amd import slater
    from temfpy_amd.engine import Engine
    from temfpy_amd.schmidt_utils import to_stopping_condition

    rng = np.random.default_rng(1952)
    L = int(rng.integers(4, 111))
    rng_h = float(rng.choice([0.7, 1.5, 3.0, 6.0]))
    cplx = bool(rng.integers(0, 2))
    x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
    H = M[0] + (1j * M[1] if cplx else 0)
    C, _ = slater.correlation_matrix(H + H.conj().T)
    C = slater.spinful_correlation_matrix(C, True)
    Lf, oc, chi = len(C), 96, 32
    assert Lf == 174
    tr = to_stopping_condition({"chi_max": chi})
    eng = Engine("cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = eng.run(C, tr, oc, Lf)
        assert eng.range_width == 128
        for (lo, hi) in ((0, 70), (70, 174)):
            part = eng.run(C, tr, oc, Lf, site_range=(lo, hi))
            assert eng.range_width == 128
            for b in range(lo, hi + 1):
                assert np.array_equal(part.bonds[b].e, full.bonds[b].e), b
                assert np.array_equal(part.bonds[b].lam, full.bonds[b].lam) and np.array_equal(part.bonds[b].masks, full.bonds[b].masks)
            for i in range(lo, hi):
                for bp, bf in zip(part.sites[i].blocks, full.sites[i].blocks):
                    assert bp[:5] == bf[:5] and np.array_equal(bp[5], bf[5]), i


def test_block_local_elimination_against_fully_pivoted_lu():
    """Default Schur-complement path (row search inside the 64 x 64 diagonal blocks, tmf_diag_inverse_batched) against the
    fully pivoted blocked LU on a chain whose always-blocks span several diagonal blocks; the forced fallback (statistics
    read back, gather repeated, fully pivoted LU) must reproduce the pivoted path bit for bit.  The block inverses stay
    O(1): that is the structure the local search relies on (the filled orbitals of neighbouring cuts pair up)."""
    from tests_inputs import random_hopping
    from temfpy_amd.engine import Engine
    from temfpy_amd.schmidt_utils import to_stopping_condition

    L, chi = 448, 160
    C, _ = orc.correlation_matrix(random_hopping(L, 2))
    tr = to_stopping_condition({"chi_max": chi})
    res = {}
    for method in ("local", "blocked", "fallback"):
        eng = Engine("cuda:0")
        eng.lu_method = method
        res[method] = eng.run(C, tr, L // 2, L)
        eng._stage_timings()
        res[method + "_info"] = (eng.kernel_info.lu_min_pivot, eng.kernel_info.lu_max_inverse, eng.kernel_info.lu_fallbacks)
    assert max(b.n_filled_left for b in res["local"].bonds) > 64      # always-blocks of several diagonal blocks
    min_pivot, max_inv, fallbacks = res["local_info"]
    assert fallbacks == 0 and 0 < max_inv < 10 and min_pivot > 0, res["local_info"]
    assert res["fallback_info"][2] == 1 and res["blocked_info"][1] == 0.0
    worst = 0.0
    for a, b, c in zip(res["local"].sites, res["blocked"].sites, res["fallback"].sites):
        assert b.det_always == c.det_always
        worst = max(worst, abs(a.det_always - b.det_always) / abs(b.det_always))
        for u, v, w in zip(a.blocks, b.blocks, c.blocks):
            assert u[:5] == v[:5] == w[:5]
            assert np.array_equal(v[5], w[5])
            worst = max(worst, np.abs(u[5] - v[5]).max() / max(np.abs(v[5]).max(), 1e-300))
    assert worst < 1e-11, worst


def test_real_dtype_path_matches_complex_path():
    """A real correlation matrix takes the float64 kernels; promoted to complex it must give the
    same Schmidt data and the same state."""
    from tests_inputs import uniform_chain

    L, chi = 40, 64
    C, _ = orc.correlation_matrix(uniform_chain(L) + np.diag(0.3 * np.cos(1.7 * np.arange(L))))
    assert not np.iscomplexobj(C)
    m_r = run_hip(C, chi)
    m_c = run_hip(C.astype(complex), chi)
    assert m_r.sites[3].blocks[0][5].dtype == np.float64 and m_c.sites[3].blocks[0][5].dtype == np.complex128
    for b in range(L + 1):
        np.testing.assert_array_equal(m_r.bonds[b].masks, m_c.bonds[b].masks)
        np.testing.assert_allclose(m_r.bonds[b].lam, m_c.bonds[b].lam, rtol=0, atol=1e-9)  # weak orbitals: e ~ 1e-11 +- 1e-16
    T1, T2 = m_r.dense_tensors(), m_c.dense_tensors()
    oc = L // 2
    ov = abs(orc.mps_overlap(T1, m_r.lam[oc], T2, m_c.lam[oc], oc))
    n1 = abs(orc.mps_overlap(T1, m_r.lam[oc], T1, m_r.lam[oc], oc))
    assert abs(ov / n1 - 1) < 1e-9
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi})
    assert abs(1 - overlap(cuts, sites, m_r, oc)) < 1e-9


def test_sector_filter_and_unlimited_chi():
    """trunc_par.sectors (schmidt_utils.py:22-32) and chi_max=None (svd_min-limited)."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater

    L = 12
    C, N = orc.correlation_matrix(random_hopping(L, 9))
    mps = slater.C_to_MPS(C, {"chi_max": None}, as_tenpy=False)
    cuts, sites = orc.c_to_mps(C, {"chi_max": None})
    for b in range(L + 1):
        np.testing.assert_array_equal(mps.bonds[b].sets, cuts[b].sets)
    assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9
    # untruncated up to svd_min: the MPS reproduces C (src/examples/slater.py:30-36)
    G = orc.mps_correlation(mps.dense_tensors(), mps.lam[L // 2], L // 2)
    np.testing.assert_allclose(G, C, atol=1e-8)
    with pytest.raises(ValueError):
        slater.C_to_MPS(C, {"chi_max": 8, "sectors": [L + 5]}, as_tenpy=False)


@pytest.mark.parametrize("L,chi", [(24, 64), (48, 96)])
def test_spinful_ph_chain_block_decoupled_spectra(L, chi):
    """BASELINE config 5 shape (uniform chain, spinful="PH") at sizes the oracle finishes: the
    two spin species make C exactly block-decoupled, every entangled eigenvalue is two-fold
    degenerate and the range-finder slabs are exactly rank deficient (regression test for the
    loss of orthogonality of Gram-Schmidt on such slabs).  Mode counts must match exactly and
    eigenvalues to 1e-11; which member of a (numerically split) degenerate Schmidt multiplet
    survives chi_max is decided by rounding in the reference as well, so S and the state are
    compared at that noise level."""
    from tests_inputs import uniform_chain
    from temfpy_amd import slater

    C, _ = orc.correlation_matrix(uniform_chain(L))
    mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False, spinful="PH")
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi}, spinful="PH")
    assert mps.L == 2 * L
    for b in range(2 * L + 1):
        c, m = cuts[b], mps.bonds[b]
        assert (c.k, c.n_filled("L"), c.n_filled("R")) == (len(m.e), m.n_filled_left, m.n_filled_right)
        np.testing.assert_allclose(m.e, c.e, rtol=0, atol=1e-11)
    dS = np.abs(orc.entropies(cuts) - mps.entanglement_entropy(all_bonds=True)).max()
    assert dS < 2e-3
    assert abs(1 - overlap(cuts, sites, mps, L)) < 1e-3


@pytest.mark.parametrize("L,ph", [(10, True), (12, False)])
def test_spinful_chain_without_chi_limit_matches_oracle_tightly(L, ph):
    """The same block-decoupled, exactly degenerate input with `chi_max=None`: the list is then cut by the svd_min
    threshold alone, which a multiplet straddles only if the threshold falls inside its 1e-7-wide noise band, so - unlike
    with a chi_max that cuts multiplets - the kept SET of patterns is determined and the comparison is sharp: chi equal,
    sorted Schmidt values 1e-10, entropies 1e-10, state overlap 1 - 1e-9 (config 5's Slater stage at oracle sizes)."""
    from tests_inputs import uniform_chain
    from temfpy_amd import slater

    C, _ = orc.correlation_matrix(uniform_chain(L) + 0.2 * np.diag(np.cos(0.9 * np.arange(L))))
    kind = "PH" if ph else "simple"
    mps = slater.C_to_MPS(C, {"chi_max": None}, as_tenpy=False, spinful=kind)
    cuts, sites = orc.c_to_mps(C, {"chi_max": None}, spinful=kind)
    assert mps.L == 2 * L
    for b in range(2 * L + 1):
        c, m = cuts[b], mps.bonds[b]
        assert (c.k, c.n_filled("L"), c.n_filled("R")) == (len(m.e), m.n_filled_left, m.n_filled_right)
        np.testing.assert_allclose(m.e, c.e, rtol=0, atol=1e-12)
        assert m.chi == len(c.lam), (b, m.chi, len(c.lam))
        np.testing.assert_allclose(np.sort(m.lam), np.sort(c.lam), rtol=0, atol=1e-10)
        assert sorted(map(bytes, np.packbits(m.sets, axis=1))) == sorted(map(bytes, np.packbits(c.sets, axis=1)))
    dS = np.abs(orc.entropies(cuts) - mps.entanglement_entropy(all_bonds=True)).max()
    assert dS < 1e-10, dS
    assert abs(1 - overlap(cuts, sites, mps, L)) < 1e-9


def test_schmidt_decomposition_self_check_on_device():
    """testing.check_schmidt_decomposition (testing.py:131-177) evaluated by tmf_recon_error_batched:
    deviations of the centre cut far below diag_tol for a proper Slater determinant, reported as
    ComparisonWarning / AssertionError / ignored according to TEST_ACTION when C is not a projector."""
    import warnings
    from tests_inputs import random_hopping
    from temfpy_amd import slater, testing

    C, _ = slater.correlation_matrix(random_hopping(32, 3))
    mps = run_hip(C, 64)
    chk = mps.info["checks"]
    assert set(chk) == {"vL is not unitary", "vL does not diagonalise C_LL", "vR is not unitary",
                        "vR does not diagonalise C_RR", "vL and vR do not SVD C_LR"}
    assert max(chk.values()) <= 1e-11, chk
    # an oracle-side cross-check of the same quantities
    e, v = np.linalg.eigh(C[:16, :16])
    assert abs(np.abs((v * e) @ v.conj().T - C[:16, :16]).max() - 0.0) <= 1e-13

    rng = np.random.default_rng(0)
    N = rng.standard_normal((32, 32)) + 1j * rng.standard_normal((32, 32))
    Cbad = C + 1e-7 * (N + N.conj().T)          # not idempotent any more: reconstructions are off by > 1e-8
    old = testing.TEST_ACTION
    try:
        testing.TEST_ACTION = "warn"
        with pytest.warns(testing.ComparisonWarning, match="does not diagonalise"):
            bad = run_hip(Cbad, 64)
        assert max(bad.info["checks"].values()) > 1e-8
        testing.TEST_ACTION = "raise"
        with pytest.raises(AssertionError, match="Max absolute difference"):
            run_hip(Cbad, 64)
        testing.TEST_ACTION = "pass"
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            quiet = run_hip(Cbad, 64)
        assert quiet.info["checks"] == {}
        # a looser diag_tol accepts it (slater.py:1216-1224 keyword)
        testing.TEST_ACTION = "raise"
        run_hip(Cbad, 64, diag_tol=1e-2)
    finally:
        testing.TEST_ACTION = old


def test_self_check_deviations_match_the_reference():
    """The device-evaluated deviations of check_schmidt_decomposition against the values the reference's
    own SchmidtModes give at the centre cut (tests/golden/ref_checks.json, make_golden_checks.py):
    the C_LR reconstruction error is dominated by the discarded modes (sigma < 1e-6) and agrees to 3
    digits - the reference warns for exactly the same inputs; everything else is at rounding level."""
    import json
    import warnings
    from tests_inputs import random_hopping
    from temfpy_amd import slater, testing

    ref = json.load(open(os.path.join(GOLDEN, "ref_checks.json")))
    for case in ref.values():
        C, _ = slater.correlation_matrix(random_hopping(case["L"], case["seed"]))
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            mps = run_hip(C, case["chi"])
        got, want = mps.info["checks"], case["deviations"]
        lr = "vL and vR do not SVD C_LR"
        if want[lr] > 1e-10:
            assert abs(got[lr] - want[lr]) <= 2e-3 * want[lr], (got[lr], want[lr])
        for key in want:
            if key != lr or want[lr] <= 1e-10:
                assert got[key] <= max(10 * want[key], 1e-9), (key, got[key], want[key])
        warned = sorted({str(w.message).strip().split("\n")[0] for w in rec
                         if isinstance(w.message, testing.ComparisonWarning)})
        assert warned == case["warned"], (warned, case["warned"])


def test_entanglement_rank_beyond_64_uses_the_wider_range_finder():
    """A dense random Hamiltonian (volume-law ground state): every orbital of the centre blocks is
    entangled (rank 80 > 64 columns), so the sweep falls back to the 128-column range finder and the
    block Jacobi kernels; parity with the oracle as everywhere else.  With the ladder cut short the
    call must refuse (NotImplementedError) instead of returning degraded orbitals."""
    from temfpy_amd import slater
    from temfpy_amd.slater import _engine

    rng = np.random.default_rng(12)
    L, chi = 160, 48
    M = rng.standard_normal((L, L)) + 1j * rng.standard_normal((L, L))
    C, N = slater.correlation_matrix(M + M.conj().T)
    mps = run_hip(C, chi)
    assert mps.info["range_finder_columns"] == 128
    assert max(len(b.e) for b in mps.bonds) > 64
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi})
    for b in range(L + 1):
        np.testing.assert_allclose(mps.bonds[b].e, cuts[b].e, rtol=0, atol=1e-13)
        np.testing.assert_allclose(mps.lam[b], cuts[b].lam, rtol=0, atol=1e-9)
    S = mps.entanglement_entropy(all_bonds=True)
    Sref = np.array([-(c.lam**2 * np.log(c.lam**2)).sum() for c in cuts])
    assert np.abs(S - Sref).max() < 1e-10
    assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9
    assert max(mps.info["checks"].values()) < 1e-10

    eng = _engine("cuda:0")
    old = eng.range_ladder
    try:
        eng.range_ladder = (64,)
        with pytest.raises(NotImplementedError, match="entanglement rank"):
            run_hip(C, chi)
    finally:
        eng.range_ladder = old


def test_config3_full_size_against_oracle_samples():
    """BASELINE config 3 at FULL size (L=1024, chi=512, the benchmark workload): the oracle is too slow
    for all 1024 sites (~5 min), so six sites spread over the chain - including both neighbours of the
    centre - are recomputed with it and compared like the small cases: occupation patterns exactly,
    e <= 1e-13, lambda <= 1e-9, S <= 1e-10, |tensor blocks| <= 1e-4 max and 3e-6 in the Frobenius norm
    (see the comment at the assertion).  Plus size-independent
    properties on every bond: normalisation, charge bookkeeping, chi profile."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater

    L, chi = 1024, 512
    C, N = slater.correlation_matrix(random_hopping(L, 0))
    oc = L // 2
    mps = run_hip(C, chi)
    assert max(mps.chi) == chi and mps.chi[0] == 1 and mps.chi[-1] == 1
    for b in range(L + 1):
        bd = mps.bonds[b]
        assert abs((bd.lam**2).sum() - 1) < 1e-12
        assert np.all(bd.q_left == bd.n_filled_left + bd.sets.sum(axis=1))
        assert bd.n_filled_left + bd.n_filled_right + len(bd.e) == N
    trunc = orc.as_trunc({"chi_max": chi})
    for i in (0, 200, oc - 1, oc, 700, L - 1):
        if i >= oc:
            bra = orc.cut_vectors(C, i + 1, trunc, "R")
            ket = orc.cut_vectors(C, i, trunc, "R" if i > oc else "LR")
            site = orc.site_tensor(bra, ket, "right")
        else:
            bra = orc.cut_vectors(C, i, trunc, "L")
            ket = orc.cut_vectors(C, i + 1, trunc, "L" if i + 1 < oc else "LR")
            site = orc.site_tensor(bra, ket, "left")
        for c in (bra, ket):
            bd = mps.bonds[c.x]
            np.testing.assert_array_equal(bd.sets, c.sets)
            np.testing.assert_allclose(bd.e, c.e, rtol=0, atol=1e-13)
            np.testing.assert_allclose(bd.lam, c.lam, rtol=0, atol=1e-9)
            p1, p2 = bd.lam**2, c.lam**2
            assert abs((p1 * np.log(p1)).sum() - (p2 * np.log(p2)).sum()) < 1e-10
        s = mps.sites[i]
        assert sorted(b[0] for b in s.blocks) == sorted(site.blocks)
        num = den = 0.0
        for q, r0, r1, c0, c1, blk in s.blocks:
            ref = site.blocks[q][4]
            assert blk.shape == ref.shape and (r0, r1, c0, c1) == tuple(site.blocks[q][:4])
            # At this size every bulk cut has eigenvalues within 0.03-0.3 decades of the 1e-12 cutoff whose
            # eigenvectors LAPACK itself resolves only to eps / gap ~ 1e-3; entries change at second order
            # in that mixing (measured with tools/diag_full_size.py: 3e-6 ... 2e-5 for the worst of ~1e5
            # entries of a site, depending on the summation order inside the GEMMs; 2e-7 in the Frobenius
            # norm, on every bulk site alike).  The Frobenius bound is the sharp one.
            np.testing.assert_allclose(np.abs(blk), np.abs(ref), rtol=0, atol=1e-4 * max(1.0, np.abs(ref).max()))
            num += ((np.abs(blk) - np.abs(ref)) ** 2).sum()
            den += (np.abs(ref) ** 2).sum()
        assert np.sqrt(num / den) < 3e-6   # site 700: 9e-7 (an eigenvalue 7 % above the cutoff), others 1e-7 ... 3e-7


@pytest.mark.parametrize("L,seed,real", [(64, 0, False), (200, 1, False), (96, 0, True)])
def test_correlation_matrix_on_device_matches_eigh(L, seed, real):
    """slater.correlation_matrix(H, device=...) - the occupied projector as (1 - sign H) / 2 by the GEMM-only
    Newton-Schulz iteration - against the host eigh path (slater.py:1150-1180): same N, entries to 1e-11
    (the projector is determined to eps * ||H|| / gap), idempotent to 1e-12."""
    from tests_inputs import random_hopping, uniform_chain
    from temfpy_amd import slater

    H = uniform_chain(L) + 0.3 * np.diag((-1.0) ** np.arange(L)) if real else random_hopping(L, seed)
    C0, N0 = slater.correlation_matrix(H)
    C1, N1 = slater.correlation_matrix(H, device="cuda:0")
    assert N1 == N0 and C1.dtype == C0.dtype
    np.testing.assert_allclose(C1, C0, rtol=0, atol=1e-11)
    np.testing.assert_allclose(C1 @ C1, C1, rtol=0, atol=1e-12)


@pytest.mark.parametrize("L,seed,N", [(64, 0, 20), (120, 2, 61), (48, 1, 1), (48, 1, 47), (32, 3, 0), (32, 3, 32)])
def test_correlation_matrix_on_device_with_particle_number(L, seed, N):
    """slater.correlation_matrix(H, N, device=...): the N lowest levels by bisection on the chemical potential with the sign
    iteration as level counter, against the host eigh path (slater.py:1174-1179); a degenerate Fermi level is refused."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater

    H = random_hopping(L, seed)
    C0, N0 = slater.correlation_matrix(H, N)
    C1, N1 = slater.correlation_matrix(H, N, device="cuda:0")
    assert N1 == N0 == N and abs(np.trace(C1).real - N) < 1e-9
    np.testing.assert_allclose(C1, C0, rtol=0, atol=1e-10)
    if 0 < N < L:
        np.testing.assert_allclose(C1 @ C1, C1, rtol=0, atol=1e-11)


def test_correlation_matrix_on_device_degenerate_fermi_level():
    from temfpy_amd import slater

    H = np.diag([-2.0, -1.0, -1.0, 0.5, 3.0]).astype(complex)
    with pytest.raises(ValueError, match="do not fill a shell"):
        slater.correlation_matrix(H, 2, device="cuda:0")
    C, N = slater.correlation_matrix(H, 3, device="cuda:0")
    np.testing.assert_allclose(C, np.diag([1.0, 1, 1, 0, 0]), rtol=0, atol=1e-12)


@pytest.mark.parametrize("svd_min,deg_tol", [(1e-4, 1e-12), (3e-7, 1e-12), (1e-6, 1e-9)])
def test_non_default_truncation_parameters(svd_min, deg_tol):
    """trunc_par beyond chi_max (schmidt_utils.py:22-54): svd_min moves the orbital cutoff svd_min^2
    (slater.py:318) and with it the range-finder threshold; degeneracy_tol the multiplet rule.
    (svd_min = 1e-8 puts the cutoff at 1e-16: the reference itself then fails its kL == kR assertion,
    slater.py:394, on eigenvalue noise.)"""
    from tests_inputs import random_hopping

    L = 64
    C, _ = orc.correlation_matrix(random_hopping(L, 4))
    par = {"chi_max": 96, "svd_min": svd_min, "degeneracy_tol": deg_tol}
    cuts, sites = orc.c_to_mps(C, par)
    from temfpy_amd import slater
    mps = slater.C_to_MPS(C, par, as_tenpy=False)
    for b in range(L + 1):
        np.testing.assert_array_equal(mps.bonds[b].sets, cuts[b].sets)
        np.testing.assert_allclose(mps.bonds[b].e, cuts[b].e, rtol=0, atol=1e-13)
        np.testing.assert_allclose(mps.bonds[b].lam, cuts[b].lam, rtol=0, atol=1e-9)
    assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9


@pytest.mark.parametrize("driver", ["sweep", "python"])
@pytest.mark.parametrize("L,seed,tol", [(48, 3, 1e-6), (32, 1, 1e-3), (24, 4, 1e-4)])
def test_centre_pairing_inside_groups_of_close_eigenvalues(L, seed, tol, driver, monkeypatch):
    """degeneracy_tol larger than the distance of eigenvalues of the centre cut (utils.py:19-96, called at slater.py:400-407):
    the reference takes the SVD of v_L^H C_LR v_R inside each group of consecutive eigenvalues closer than the tolerance -
    an ABSOLUTE distance, so all orbitals with 1 - e below it form one group - which reorders the orbitals by descending
    sqrt(e (1 - e)) while e keeps its order: a different state than with the default 1e-12 (2e-7 ... 1e-3 in overlap for
    these inputs).  The device path follows: the oracle's state to 1e-9 with the tolerance, measurably not the same
    without it (the inputs do exercise the rule)."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater
    from temfpy_amd.engine import Engine
    if driver == "python":
        monkeypatch.setattr(Engine, "sweep_impl", "python")
    C, _ = orc.correlation_matrix(random_hopping(L, seed))
    wide, tight = {"chi_max": 300, "degeneracy_tol": tol}, {"chi_max": 300, "degeneracy_tol": 1e-12}
    cuts, sites = orc.c_to_mps(C, dict(wide))
    mps = slater.C_to_MPS(C, dict(wide), as_tenpy=False)
    for b in range(L + 1):
        np.testing.assert_array_equal(mps.bonds[b].sets, cuts[b].sets)
        np.testing.assert_allclose(mps.bonds[b].e, cuts[b].e, rtol=0, atol=1e-13)
    assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9
    other = slater.C_to_MPS(C, dict(tight), as_tenpy=False)
    assert abs(1 - overlap(cuts, sites, other, L // 2)) > 1e-7


@pytest.mark.parametrize("driver", ["cpp", "python"])
def test_general_determinant_path_through_global_memory(driver):
    """Sites whose sometimes-matrix does not fit the LDS stage of the determinant kernels (> 64 x 64 complex) or whose
    minors have more than 64 rows used to raise; they now take class 255 of tmf_det_gather_batched (matrix read from global
    memory, minor in LDS).  Such sites need ~100 strongly entangled orbitals, so the dispatch is exercised with the test
    switch TMF_DET_GLOBAL=1, which sends EVERY charge sector there: same MPS as the oracle's."""
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np
        from oracle import slater_oracle as orc
        from tests_inputs import random_hopping
        from temfpy_amd import slater
        from test_gpu_sweep import overlap
        L = 40
        C, _ = orc.correlation_matrix(random_hopping(L, 6))
        cuts, sites = orc.c_to_mps(C, {"chi_max": 48})
        mps = slater.C_to_MPS(C, {"chi_max": 48}, as_tenpy=False)
        for b in range(L + 1):
            assert np.array_equal(mps.bonds[b].sets, cuts[b].sets)
        assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9
        print("ok")
    """ % (root, os.path.join(root, "tests")))
    env = dict(os.environ, TMF_DET_GLOBAL="1", TMF_DET_METHOD="reduced", TMF_SWEEP=driver)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("case", ["L2", "empty", "chi1", "L1", "single_particle", "two_filled"])
def test_edge_cases_match_oracle(case):
    """Degenerate inputs the reference accepts: the smallest chains, an empty band, chi_max = 1, one
    particle, a chain with exactly filled sites at its ends.  (Exact product states with several filled
    sites inside one block are NOT among them: the reference's `inv(O_aa)`, slater.py:1079/1086, is
    singular whenever LAPACK orders the degenerate filled orbitals unfavourably - `numpy.linalg.LinAlgError`
    in the oracle as well.)"""
    rng = np.random.default_rng(8)
    chi = 16
    if case == "two_filled":     # filled site, generic 6-site segment, empty site
        M = rng.standard_normal((6, 6)) + 1j * rng.standard_normal((6, 6))
        Cm, _ = orc.correlation_matrix(M + M.conj().T)
        C = np.zeros((8, 8), complex)
        C[0, 0] = 1.0
        C[1:7, 1:7] = Cm
    elif case == "L2":
        H = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
        C, _ = orc.correlation_matrix(H + H.conj().T, N=1)
    elif case == "L1":
        C = np.array([[1.0]])
    elif case == "empty":
        C = np.zeros((5, 5))
    elif case == "single_particle":
        v = rng.standard_normal(9) + 1j * rng.standard_normal(9)
        v /= np.linalg.norm(v)
        C = np.outer(v, v.conj())
    else:
        M = rng.standard_normal((12, 12)) + 1j * rng.standard_normal((12, 12))
        C, _ = orc.correlation_matrix(M + M.conj().T)
        chi = 1
    L = len(C)
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi})
    mps = run_hip(C, chi)
    assert mps.L == L
    for b in range(L + 1):
        assert mps.bonds[b].chi == len(cuts[b].lam)
        np.testing.assert_allclose(mps.bonds[b].e, cuts[b].e, rtol=0, atol=1e-13)
        np.testing.assert_allclose(mps.lam[b], cuts[b].lam, rtol=0, atol=1e-10)
        assert (mps.bonds[b].n_filled_left, mps.bonds[b].n_filled_right) == (cuts[b].nfL if cuts[b].nfL is not None else mps.bonds[b].n_filled_left,
                                                                             cuts[b].nfR if cuts[b].nfR is not None else mps.bonds[b].n_filled_right)
    if L > 1:
        assert abs(1 - overlap(cuts, sites, mps, L // 2)) < 1e-9


def test_singular_always_block_raises_like_the_reference():
    """A product state with two filled sites whose overlap block is singular: numpy.linalg.LinAlgError from
    `inv(O_aa)` in the reference (slater.py:1079/1086) - or a valid result when the arbitrary basis of the
    degenerate filled orbitals happens to be favourable.  Never silent garbage."""
    C = np.diag(np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 0], float))
    try:
        mps = run_hip(C, 8)
    except np.linalg.LinAlgError:
        return
    for s in mps.sites:
        for blk in s.blocks:
            assert np.all(np.isfinite(blk[5]))
    assert all(abs((b.lam**2).sum() - 1) < 1e-12 for b in mps.bonds)


def test_chain_longer_than_the_slab_kernel_limit():
    """L = 1100 complex: the m x 64 slabs of the second range-finder QR exceed the 1024 rows the Householder slab
    kernel keeps in registers, so that QR falls back to the Gram-Schmidt path while the first one (n <= 550) stays
    on the slab kernel; entropies against the oracle on sampled cuts."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater

    L, chi = 1100, 64
    C, _ = slater.correlation_matrix(random_hopping(L, 4))
    mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False)
    S = mps.entanglement_entropy(all_bonds=True)
    trunc = orc.as_trunc({"chi_max": chi})
    for x in (3, 400, 550, 1097):
        cut = orc.cut_vectors(C, x, trunc, "LR" if x == L // 2 else ("L" if x < L // 2 else "R"))
        p = cut.lam**2
        assert abs(S[x] + (p[p > 0] * np.log(p[p > 0])).sum()) < 1e-10
        assert mps.bonds[x].chi == len(cut.lam)
