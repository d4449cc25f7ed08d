"""Full-size parity against summaries produced by the REFERENCE's own NumPy core
(tests/golden/make_golden_summary.py, run once in the build container): every bond and every site of the
BASELINE configurations, not a sample.

What is compared is what no gauge choice can change (eigenvectors are defined up to a phase, so tensors agree
up to a diagonal unitary per bond): integers exactly (chi, occupation patterns and charges through SHA-1 of
the packed arrays, filled-orbital counts, charge-block lists); e <= 1e-13; normalised Schmidt values <= 1e-9;
S(b) <= 1e-10; the norm of the unnormalised Schmidt vector to 1e-9 relative; per site the Frobenius norm of
every charge block and the 2-norm of every merged (p, bra) row.  Tolerances of the tensor norms: 1e-9 relative
at L = 256; at L = 1024 every bulk cut has eigenvalues within 0.03 - 0.3 decades of the 1e-12 orbital cutoff,
whose eigenvectors LAPACK itself resolves only to eps / gap, and entries move at second order in that mixing
(measured in round 1: 2e-7 in the Frobenius norm on every bulk site alike) - 5e-6 relative there."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FULL = os.path.join(GOLDEN, "full")


def sha(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def convert(name, **kw):
    from tests_inputs import random_hopping, uniform_chain
    from temfpy_amd import slater

    H = {"cfg2_rand_L256_s0_chi128": lambda: random_hopping(256, 0), "cfg3_rand_L1024_s0_chi512": lambda: random_hopping(1024, 0),
         "rand_L1024_s1_chi512": lambda: random_hopping(1024, 1), "cfg5_chainPH_L512_chi512": lambda: uniform_chain(512)}[name]()
    ref = np.load(os.path.join(FULL, name + ".npz"))
    C, N = slater.correlation_matrix(H)
    assert N == int(ref["N"])
    if "kw_spinful" in ref:
        kw["spinful"] = str(ref["kw_spinful"])
    return slater.C_to_MPS(C, {"chi_max": int(ref["chi_max"])}, as_tenpy=False, **kw), ref


def compare_bonds(mps, ref, e_tol=1e-13, lam_tol=1e-9, S_tol=1e-10):
    L = int(ref["L"])
    assert mps.L == L and mps.ortho_center == int(ref["ortho_center"])
    np.testing.assert_array_equal(mps.chi, ref["chi"])
    S = mps.entanglement_entropy(all_bonds=True)
    worst = dict(e=0.0, lam=0.0)
    for b in range(L + 1):
        bd = mps.bonds[b]
        assert [bd.n_filled_left, bd.n_filled_right] == ref["n_filled"][b].tolist(), b
        assert np.array_equal(sha(np.packbits(bd.sets, axis=1, bitorder="little")), ref["sets_sha1"][b]), f"sets of bond {b}"
        assert np.array_equal(sha(np.asarray(bd.q_left, np.int64)), ref["q_sha1"][b]), f"charges of bond {b}"
        e_ref = ref["e"][ref["e_off"][b]: ref["e_off"][b + 1]]
        lam_ref = ref["lam"][ref["lam_off"][b]: ref["lam_off"][b + 1]]
        worst["e"] = max(worst["e"], np.abs(bd.e - e_ref).max(initial=0.0))
        worst["lam"] = max(worst["lam"], np.abs(bd.lam - lam_ref).max())
        assert abs(np.linalg.norm(bd.lam_raw) / ref["lam_norm"][b] - 1) < 1e-9, b
    assert worst["e"] <= e_tol and worst["lam"] <= lam_tol, worst
    dS = np.abs(S - ref["S"]).max()
    assert dS <= S_tol, dS
    return dS, worst


def compare_sites(mps, ref, rtol):
    L = int(ref["L"])
    worst_blk = worst_row = 0.0
    for i in range(L):
        s = mps.sites[i]
        q_ref = ref["blk_q"][ref["blk_off"][i]: ref["blk_off"][i + 1]]
        n_ref = ref["blk_norm"][ref["blk_off"][i]: ref["blk_off"][i + 1]]
        mine = {b[0]: b for b in s.blocks}
        assert sorted(mine) == sorted(q_ref.tolist()), f"charge blocks of site {i}"
        rows = np.zeros(2 * s.chi_bra)
        for q, nr in zip(q_ref.tolist(), n_ref):
            _, r0, r1, c0, c1, blk = mine[q]
            a2 = np.abs(blk) ** 2
            worst_blk = max(worst_blk, abs(np.sqrt(a2.sum()) - nr) / max(nr, 1e-300))
            rows[r0:r1] += a2.sum(axis=1)
        r_ref = ref["row_norm"][ref["row_off"][i]: ref["row_off"][i + 1]]
        assert len(r_ref) == len(rows)
        worst_row = max(worst_row, np.abs(np.sqrt(rows) - r_ref).max() / max(r_ref.max(), 1e-300))
    assert worst_blk <= rtol, worst_blk
    assert worst_row <= max(rtol, 2e-7), worst_row     # the reference rows are stored in float32
    return worst_blk, worst_row


def test_config2_every_bond_and_site_against_the_reference():
    mps, ref = convert("cfg2_rand_L256_s0_chi128")
    dS, worst = compare_bonds(mps, ref)
    wb, wr = compare_sites(mps, ref, rtol=1e-9)
    print(f"cfg2: max|dS| {dS:.1e}, e {worst['e']:.1e}, lam {worst['lam']:.1e}, block norms {wb:.1e}, row norms {wr:.1e}")


@pytest.mark.parametrize("name", ["cfg3_rand_L1024_s0_chi512", "rand_L1024_s1_chi512"])
def test_config3_every_bond_and_site_against_the_reference(name):
    if not os.path.exists(os.path.join(FULL, name + ".npz")):
        pytest.skip("summary not generated")
    mps, ref = convert(name)
    dS, worst = compare_bonds(mps, ref)
    wb, wr = compare_sites(mps, ref, rtol=5e-6)
    print(f"{name}: max|dS| {dS:.1e}, e {worst['e']:.1e}, lam {worst['lam']:.1e}, block norms {wb:.1e}, row norms {wr:.1e}")


def test_config5_slater_stage_against_the_reference():
    """Uniform chain, spinful "PH" (1024 real-dtype MPS sites): both spin species give the same spectrum, so every
    Schmidt value comes in exactly degenerate multiplets that rounding splits at the 1e-16 level - in the reference
    too.  Which members survive `chi_max` is then decided by noise, so occupation patterns are NOT compared; the
    orbital data (counts, eigenvalues) must agree to 1e-11, chi to the size of one multiplet, and entropies to the
    weight a differently cut multiplet can carry (bounded below from the reference's own smallest kept value)."""
    mps, ref = convert("cfg5_chainPH_L512_chi512")
    L = int(ref["L"])
    S = mps.entanglement_entropy(all_bonds=True)
    worst_e = worst_S = 0.0
    for b in range(L + 1):
        bd = mps.bonds[b]
        assert [bd.n_filled_left, bd.n_filled_right] == ref["n_filled"][b].tolist(), b
        e_ref = ref["e"][ref["e_off"][b]: ref["e_off"][b + 1]]
        assert len(bd.e) == len(e_ref), b
        worst_e = max(worst_e, np.abs(bd.e - e_ref).max(initial=0.0))
        lam_ref = ref["lam"][ref["lam_off"][b]: ref["lam_off"][b + 1]]
        assert abs(bd.chi - len(lam_ref)) <= 8, (b, bd.chi, len(lam_ref))
        # a multiplet cut differently moves at most ~8 values of the size of the smallest kept one
        w = 8 * float(lam_ref.min()) ** 2
        bound = 1e-10 + w * (1 + abs(np.log(max(w, 1e-300))))
        assert abs(S[b] - ref["S"][b]) <= bound, (b, S[b], ref["S"][b], bound)
        worst_S = max(worst_S, abs(S[b] - ref["S"][b]))
        n = min(bd.chi, len(lam_ref)) - 8
        if n > 0:   # the leading values are untouched by the cut (up to the normalisation, ~w)
            np.testing.assert_allclose(np.sort(bd.lam)[::-1][:n], np.sort(lam_ref)[::-1][:n], rtol=0, atol=1e-9 + w)
    assert worst_e <= 1e-11, worst_e
    print(f"cfg5 Slater stage: max|de| {worst_e:.1e}, max|dS| {worst_S:.1e}")


def test_asynchronous_download_gives_the_same_result():
    """download="async" (bench.py's host -> host loop): two conversions in flight, tensors land in page-locked
    memory under the next conversion's kernels; both must equal the blocking call bit for bit."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater
    from temfpy_amd.engine import Engine
    from temfpy_amd.schmidt_utils import to_stopping_condition

    L = 96
    tr = to_stopping_condition({"chi_max": 64})
    Cs = [slater.correlation_matrix(random_hopping(L, s))[0] for s in (0, 1, 2)]
    eng = Engine("cuda:0")
    sync = [eng.run(C, tr, L // 2, L) for C in Cs]
    pending = [eng.run(C, tr, L // 2, L, download="async") for C in Cs]     # three results in flight
    for ref, got in zip(sync, pending):
        got.wait()
        for b in range(L + 1):
            assert np.array_equal(ref.bonds[b].lam_raw, got.bonds[b].lam_raw)
            assert np.array_equal(ref.bonds[b].masks, got.bonds[b].masks)
        for i in range(L):
            assert ref.sites[i].det_always == got.sites[i].det_always
            for x, y in zip(ref.sites[i].blocks, got.sites[i].blocks):
                assert x[:5] == y[:5] and np.array_equal(x[5], y[5])


def test_sharded_range_finder_decisions_are_global():
    """A cut on a shard boundary is computed by both neighbouring ranks; the adaptive decisions of the entangled
    stage (subspace iteration, wider range finder) change its orbitals by a gauge, so they must be taken on the
    maximum over ALL ranks.  Here the tolerance is set so that only the cuts in the middle of the chain ask for
    the subspace iteration: a shard at the end of the chain would not iterate on its own.  With the reduced
    decision every shard reproduces the unsharded result bit for bit; without it the end shard differs."""
    from tests_inputs import random_hopping
    from temfpy_amd import slater
    from temfpy_amd.engine import Engine
    from temfpy_amd.multi_gpu import shard_sites
    from temfpy_amd.schmidt_utils import to_stopping_condition

    L = 192
    C, _ = slater.correlation_matrix(random_hopping(L, 7))
    tr = to_stopping_condition({"chi_max": 32})

    class Recorder:      # the unsharded run sees every cut: its local values ARE the global maxima
        def __init__(self):
            self.seen = []

        def max(self, v):
            self.seen.append(np.array(v))
            return v

    class Replay:
        def __init__(self, seen):
            self.seen, self.i, self.local = seen, 0, []

        def max(self, v):
            self.local.append(np.array(v))
            out = np.maximum(v, self.seen[self.i])
            self.i += 1
            return out

    eng = Engine("cuda:0")
    eng.run(C, tr, L // 2, L)
    assert eng.range_floor > 0
    eng.range_floor_tol = 0.5 * eng.range_floor          # the middle cuts now ask for one subspace iteration
    rec = Recorder()
    eng.coord = rec
    full = eng.run(C, tr, L // 2, L)
    assert eng.range_iterations_used == 1 and len(rec.seen) == 2
    ranges = shard_sites(L, L // 2, 4)
    alone = []
    for lo, hi in ranges:
        rp = Replay(rec.seen)
        eng.coord = rp
        part = eng.run(C, tr, L // 2, L, site_range=(lo, hi))
        assert rp.i == len(rec.seen) and eng.range_iterations_used == 1
        alone.append(rp.local[0][0] <= eng.range_floor_tol)      # would this shard have skipped the iteration?
        for i in range(lo, hi):
            for x, y in zip(part.sites[i].blocks, full.sites[i].blocks):
                assert x[:5] == y[:5] and np.array_equal(x[5], y[5])
        for b in range(lo, hi + 1):
            assert np.array_equal(part.bonds[b].lam_raw, full.bonds[b].lam_raw)
    assert alone[0] and alone[-1] and not all(alone)     # the hazard is real on this input
    eng.coord = None
    part = eng.run(C, tr, L // 2, L, site_range=ranges[0])       # undecorated end shard: no iteration, other gauge
    assert eng.range_iterations_used == 0
