"""The read-only views with the reference's dataclass field names (temfpy_amd/views.py) against the fields of the
REFERENCE's own objects (tests/golden/views_*.npz, made by tests/golden/make_golden_views.py).  CPU only: the bond /
site data the views wrap is produced here by the oracle (pinned to the reference by tests/test_oracle_golden.py)."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import slater_oracle as orc
from temfpy_amd.mps_data import BondData, MPSData, SiteData


def _mps_from_oracle(C, chi, oc):
    cuts, sites = orc.c_to_mps(C, {"chi_max": chi}, ortho_center=oc)
    bonds = []
    for b, c in enumerate(cuts):
        m = np.zeros((len(c.sets), 2), np.uint64)
        for i in range(c.k):
            m[:, i // 64] |= c.sets[:, i].astype(np.uint64) << np.uint64(i % 64)
        bonds.append(BondData(x=b, e=c.e, n_filled_left=c.n_filled("L"), n_filled_right=c.n_filled("R"), masks=m,
                              lam_raw=c.lam_raw, lam=c.lam, q_left=c.n_filled("L") + c.sets.sum(axis=1)))
    sds = []
    for i, s in enumerate(sites):
        blocks = [(q, r0, r1, c0, c1, blk) for q, (r0, r1, c0, c1, blk) in s.blocks.items()]
        bra = bonds[i] if s.mode == "left" else bonds[i + 1]
        ket = bonds[i + 1] if s.mode == "left" else bonds[i]
        sds.append(SiteData(mode=s.mode, det_always=s.det_always, qtotal=0, bra_p=s.bra_p, bra_alpha=s.bra_alpha, blocks=blocks,
                            chi_bra=bra.chi, chi_ket=ket.chi))
    return MPSData(bonds, sds, oc, len(C))


def test_views_match_the_reference_objects():
    g = np.load(os.path.join(GOLDEN, "views_rand_L14_s6_chi24.npz"))
    L, oc = int(g["L"]), int(g["ortho_center"])
    mps = _mps_from_oracle(g["C_in"], int(g["chi_max"]), oc)
    keys = ("filled", "entangled", "empty")
    for b in range(L + 1):
        V = mps.schmidt_vectors(b)
        m = mps.schmidt_modes(b)
        assert [m.n_filled("L"), m.n_filled("R"), m.n_fermion, m.nL, m.nR] == g[f"b{b}_nf"].tolist()
        np.testing.assert_allclose(V.schmidt_values, g[f"b{b}_sv"], rtol=1e-12)
        assert V.n_schmidt == len(g[f"b{b}_sv"]) and V.n_entangled == m.n_entangled
        for side, ix in (("L", m.ixL), ("R", m.ixR)):
            if f"b{b}_ix{side}" in g:      # the reference computes that side at this cut
                assert [[ix[k].start, ix[k].stop] for k in keys] == g[f"b{b}_ix{side}"].tolist()
                np.testing.assert_allclose(m.eigenvalues(side), g[f"b{b}_eig{side}"], atol=1e-12)
                np.testing.assert_array_equal(V.sets(side), g[f"b{b}_{'left' if side == 'L' else 'right'}_sets"])
        if f"b{b}_singular_values" in g:
            np.testing.assert_allclose(m.singular_values, g[f"b{b}_singular_values"], atol=1e-12)
        np.testing.assert_allclose(m.schmidt_values(mps.bonds[b].sets), g[f"b{b}_sv"], rtol=1e-12)
    for i in range(L):
        T = mps.tensor_data(i)
        assert T.mode == ("left" if i < oc else "right") and T.physical_leg and T.qtotal == int(g[f"s{i}_qtotal"])
        assert [[q, s.start, s.stop] for q, s in T.idx_bra.items()] == g[f"s{i}_idx_bra"].tolist()
        assert [[q, s.start, s.stop] for q, s in T.idx_ket.items()] == g[f"s{i}_idx_ket"].tolist()
        # slater.py:1133-1141: the block of ket charge q sits in the bra rows of charge q + qtotal * qconj
        for q, blk in T.blocks.items():
            sl_b, sl_k = T.merged_idx[q], T.idx_ket[q]
            assert blk.shape == (sl_b.stop - sl_b.start, sl_k.stop - sl_k.start)
        # the merged leg holds, per charge Q, the p = 0 rows of bra sector Q then the p = 1 rows of the neighbouring sector
        n_rows = sum(sl.stop - sl.start for sl in T.merged_idx.values())
        assert n_rows == 2 * sum(sl.stop - sl.start for sl in T.idx_bra.values())
