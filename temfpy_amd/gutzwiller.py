"""Gutzwiller projections of a finite or infinite fermionic MPS to a spin-1/2 chain (temfpy/gutzwiller.py), on MI355X.

``abrikosov`` (gutzwiller.py:95-281) and ``abrikosov_ph`` (:284-486) keep the reference's names, keyword
arguments, defaults, warnings and exceptions.  They take the result of ``slater.C_to_MPS`` /
``pfaffian.C_to_MPS`` (``MPSData`` / ``PfMPSData``) and return a :class:`SpinMPSData`: the right-canonical
MPS of the normalised projected state with its Schmidt values and (for number-conserving input of
``abrikosov_ph``) 2 S^z charges.

Everything numerical runs on the GPU through the C ABI (``include/temfpy_hip.h``); there is no CPU path:

* pair contraction + projection (TeNPy ``group_sites(2)`` + ``iproject``, gutzwiller.py:227-244, :409-444):
  the charge blocks of both fermion tensors are regrouped by ``tmf_copy_blocks_batched`` and multiplied by ONE
  batched MFMA launch (``tmf_gemm_batched``) over all pairs, physical states and charge sectors;
* canonical form (TeNPy ``MPS.canonical_form_finite(cutoff=...)``, called at gutzwiller.py:266 / :471).
  ``method="parallel"`` (default): two QR-only sweeps that do not depend on each other run concurrently on two HIP
  streams (rightwards ``R_j T_j = A_j R_{j+1}``, leftwards ``T_j L_{j+1} = L_j B_j``; ``tmf_house_slab_batched``,
  Householder: the projected tensors are exactly rank deficient, where Gram-Schmidt needs a rank decision and
  Householder does not), the centre matrix of every bond is ``C_j = R_j L_j / norm``, the SVDs of ALL bonds and charge
  sectors are ONE Jacobi launch (~2500 workgroups instead of <= 5 at a time), the Schmidt gauge
  ``B_j <- V_j^H B_j V_{j+1}`` two batched MFMA launches, and one batched Gram-Schmidt pass over the kept rows of
  every ``B_j`` makes the tensors exactly right-isometric again (every bond is truncated on its own, which costs a row
  with Schmidt value s up to (cutoff / s)^2 of its norm; the correction moves the state by less than the truncation).
  For the same reason Schmidt values within a decade of ``cutoff`` come out up to ~cutoff larger than TeNPy's, which
  truncates the bonds to the right of a site before it takes that site's SVD; all others agree to 1e-15.
  ``method="sequential"`` is TeNPy's algorithm step by step: a sweep to the right with one QR per charge block, then a
  sweep back with one SVD per site, N = A X of shape chi_l x (2 chi_r) per charge block: QR of N^H, a second QR of the
  small factor R^H as preconditioner, one-sided Jacobi WITHOUT accumulator on R3^H
  (``tmf_jacobi_compact_batched``: its normalised columns are the right singular vectors V3 of R3), B = (Q V3)^H,
  U S = N (Q V3) pushed to the left.  Same state, norm and Schmidt values (1e-12), 3.4x slower at L = 512, chi = 512:
  every step waits for the previous one and occupies <= 5 of 256 CUs.
  In both methods Schmidt values below ``cutoff`` are zeroed on the device (shapes stay fixed: all descriptors
  are built once, no host round trip inside a sweep) and compacted on the host.

Infinite MPS (gutzwiller.py:197-206, ``canonical_form_infinite1`` at :272 / :475): the unit cell ``iMPS.iMPSData`` returned by
``C_to_iMPS`` / ``H_to_iMPS`` is projected with the reference's masks (``q_left`` required for ``abrikosov``; ``parity`` and
``offset`` for ``abrikosov_ph``) and returned as :class:`SpiniMPSData`.  The cell is closed by the dominant eigenvectors of
its transfer matrix (Arnoldi: Krylov bookkeeping on the host, every application of the cell on the GPU), a boundary gauge
from two Jacobi launches, and then canonicalised by the finite algorithm between the fixed boundaries (``_close_cell``).
For finite input ``q_left`` / ``offset`` / ``parity`` reduce to the warnings the reference emits.
"""
from __future__ import annotations

import logging
import os
import time
from typing import Literal
from warnings import warn

import numpy as np

from . import _native as nat

logger = logging.getLogger(__name__)

# Infinite MPS: the fixed points of the transfer matrix are Gram matrices, so Schmidt values below sqrt(machine epsilon)
# are rounding noise (TeNPy's canonical_form_infinite1 drops Gram eigenvalues below 2 eps for the same reason).
_INF_GRAM_EPS = 1e-15
_INF_SCHMIDT_EPS = 3e-8


# ---------------------------------------------------------------------------------------------------
# result container
# ---------------------------------------------------------------------------------------------------
class SpinMPSData:
    """Finite spin-1/2 MPS in right-canonical form (``form = ['B'] * L``), p = 0: down, p = 1: up.

    ``blocks[j]``: list of (p, q_l, q_r, l0, l1, r0, r1, ndarray (l1-l0, r1-r0)) charge blocks of B_j;
    ``lam[b]``: normalised Schmidt values of bond b (ordered by charge, inside a charge descending);
    ``charges[b]``: 2 S^z to the left of bond b per Schmidt index (``conserve == 'Sz'``) or the retained
    fermionic sector label (``conserve is None``; TeNPy drops it, gutzwiller.py:244 / :444);
    ``norm``: norm of the projected state before normalisation (the weight TeNPy discards with
    ``renormalize=True``)."""

    def __init__(self, blocks, lam, charges, conserve, norm, unit_cell_width, canonical=True, timings=None):
        self.blocks, self._lam, self.charges = blocks, lam, charges
        self.conserve, self.norm = conserve, norm
        self.L = len(blocks)
        self.unit_cell_width = unit_cell_width
        self.form = ["B"] * self.L if canonical else [None] * self.L
        self.timings = timings or {}

    @property
    def lam(self):
        return self._lam

    @property
    def chi(self):
        return [len(x) for x in self._lam]

    def entanglement_entropy(self, all_bonds=False):
        out = np.zeros(self.L + 1)
        for i, s in enumerate(self._lam):
            p = np.asarray(s) ** 2
            p = p[p > 0]
            out[i] = -(p * np.log(p)).sum()
        return out if all_bonds else out[1:-1]

    def dense_tensors(self):
        out = []
        for j, bl in enumerate(self.blocks):
            dt = bl[0][7].dtype if bl else float
            T = np.zeros((2, len(self._lam[j]), len(self._lam[j + 1])), dt)
            for p, _, _, l0, l1, r0, r1, a in bl:
                T[p, l0:l1, r0:r1] = a
            out.append(T)
        return out

    def to_tenpy(self):
        """``tenpy.networks.mps.MPS`` on ``SpinHalfSite(conserve)`` (gutzwiller.py:403, :449-451).
        Needs physics-tenpy, which is not installed in the build environment: untested there."""
        import tenpy.linalg.np_conserved as npc
        from tenpy import networks

        site = networks.site.SpinHalfSite("Sz" if self.conserve == "Sz" else None)
        Bs = []
        for j, T in enumerate(self.dense_tensors()):
            if self.conserve == "Sz":
                legs = [npc.LegCharge.from_qflat(site.leg.chinfo, [[q] for q in self.charges[j]], qconj=+1),
                        site.leg,
                        npc.LegCharge.from_qflat(site.leg.chinfo, [[q] for q in self.charges[j + 1]], qconj=-1)]
                # site.leg is sorted by charge: [down (-1), up (+1)] = our p order
                B = npc.Array.from_ndarray(T.transpose(1, 0, 2), legs, labels=["vL", "p", "vR"], cutoff=0.0)
            else:
                B = npc.Array.from_ndarray_trivial(T.transpose(1, 0, 2), labels=["vL", "p", "vR"])
            Bs.append(B)
        psi = networks.mps.MPS([site] * self.L, Bs, self._lam, form="B", unit_cell_width=self.unit_cell_width)
        psi._temfpy_amd = self
        return psi


class SpiniMPSData(SpinMPSData):
    """Unit cell of an infinite spin-1/2 MPS in right-canonical form (the ``bc='infinite'`` result of gutzwiller.py:272 /
    :475).  As :class:`SpinMPSData`, with ``lam[L] == lam[0]`` and ``charges[L] == charges[0]``; the blocks of the last
    site obey q_l + 2 S^z(p) = q_r + ``cell_charge`` (2 S^z of one cell), as in ``iMPS.iMPSData``.  ``norm`` is the norm of
    the projected state per unit cell (square root of the dominant transfer-matrix eigenvalue)."""

    bc = "infinite"

    def __init__(self, blocks, lam, charges, conserve, norm, unit_cell_width, cell_charge, timings=None):
        super().__init__(blocks, lam, charges, conserve, norm, unit_cell_width, True, timings)
        self.cell_charge = cell_charge

    def to_tenpy(self):
        """``tenpy.networks.mps.MPS(..., bc="infinite", form="B")`` on ``SpinHalfSite`` (what gutzwiller.py:272 / :475 leave
        behind).  Needs physics-tenpy: written against its documented interface, exercised by no test here."""
        from tenpy import networks

        from .iMPS import cell_to_tenpy

        return cell_to_tenpy(self, networks.site.SpinHalfSite("Sz" if self.conserve == "Sz" else None),
                             charged=self.conserve == "Sz")


# ---------------------------------------------------------------------------------------------------
# input adapters: charge blocks of the fermion tensors as column-major matrices inside one flat buffer
# ---------------------------------------------------------------------------------------------------
_blk_dt = np.dtype([("site", "<i4"), ("p", "<i4"), ("cl", "<i4"), ("cr", "<i4"), ("off", "<i8"), ("ld", "<i4"),
                    ("rows", "<i4"), ("cols", "<i4"), ("trans", "<i4")])


class _Fermions:
    """L, dtype, charges[b] (ascending ints per Schmidt index), lam_c, oc, conserve, flat (torch tensor or
    ndarray), blocks (_blk_dt records): B^p_{cl -> cr} = src (trans 0) or src^T (trans 1), src = rows x cols
    column-major with leading dimension ld at element offset `off` of flat."""


def _sector_table(q):
    """{charge: (start, n)} of an ascending per-index charge array."""
    q = np.asarray(q)
    if q.size == 0:
        return {}
    assert np.all(np.diff(q) >= 0), "bond charges must be sorted"
    vals, start = np.unique(q, return_index=True)
    stop = np.concatenate((start[1:], [q.size]))
    return {int(v): (int(a), int(b - a)) for v, a, b in zip(vals, start, stop)}


def _sector_tables(charges):
    """_sector_table of every bond from one pass over the concatenated charges."""
    nb = len(charges)
    sizes = np.array([len(q) for q in charges], np.int64)
    if nb == 0 or sizes.sum() == 0:
        return [{} for _ in range(nb)]
    q = np.concatenate([np.asarray(x, np.int64) for x in charges])
    bond = np.repeat(np.arange(nb), sizes)
    off = np.concatenate(([0], np.cumsum(sizes)))
    same_bond = bond[1:] == bond[:-1]
    assert np.all(q[1:][same_bond] >= q[:-1][same_bond]), "bond charges must be sorted"
    new = np.concatenate(([True], ~same_bond | (q[1:] != q[:-1])))
    st = np.nonzero(new)[0]                                   # first index of every (bond, charge) run
    ln = np.diff(np.concatenate((st, [q.size])))
    rb = bond[st]
    first_run = np.searchsorted(rb, np.arange(nb + 1))
    v_l, s_l, n_l = q[st].tolist(), (st - off[rb]).tolist(), ln.tolist()
    return [dict(zip(v_l[a:b], zip(s_l[a:b], n_l[a:b]))) for a, b in zip(first_run[:-1].tolist(), first_run[1:].tolist())]


def _fermions_from_shard(mps):
    """The same adapter from the flat tables of a conversion that one rank did in full (the usual case): every record of
    every site at once, no per-site objects (87 -> ~10 ms of host time at 1024 sites).  None if the MPS is not of that kind."""
    shards, flat_t = getattr(mps, "shards", None), getattr(mps, "_flat_t", None)
    if not shards or len(shards) != 1 or flat_t is None:
        return None
    sh = shards[0]
    A, L = sh.arrays, mps.L
    if int(sh.meta["s_lo"]) != 0 or int(sh.meta["s_hi"]) != L or "out" not in A or A["out"].size == 0:
        return None
    if sh.wait is not None:
        sh.wait()
    base = flat_t.numpy()
    if base.__array_interface__["data"][0] != A["out"].__array_interface__["data"][0] or base.dtype != A["out"].dtype:
        return None
    f = _Fermions()
    f.L, f.oc, f.conserve, f.perm = L, mps.ortho_center, "N", None
    cpos = sh.cpos
    chi = A["c_chi"][cpos[np.arange(L + 1)]].astype(np.int64)
    f.charges = [np.asarray(A["c_q"][cpos[b], : chi[b]], np.int64) for b in range(L + 1)]
    f.tabs = _sector_tables(f.charges)
    f.lam_c = np.asarray(mps.bonds[mps.ortho_center].lam)
    nsec = A["nsec"].astype(np.int64)
    site = np.repeat(np.arange(L), nsec)
    first = np.concatenate(([0], np.cumsum(nsec)))[:-1]
    sec = A["sectors"][np.repeat(A["sec_off"].astype(np.int64), nsec) + (np.arange(int(nsec.sum())) - first[site])]
    r0, r1 = sec["r0"].astype(np.int64), sec["r1"].astype(np.int64)
    ncol = (sec["c1"] - sec["c0"]).astype(np.int64)
    bo = A["bra_off"].astype(np.int64)[site]
    # rows of a sector: its p = 0 run, then its p = 1 run (stable sort of the merged leg by particle number)
    cp = np.concatenate(([0], np.cumsum(np.asarray(A["bra_p"], np.int64))))
    ones = cp[bo + r1] - cp[bo + r0]
    k0 = (r1 - r0) - ones
    left = (A["mode"][site] == 0)
    chain_off = np.concatenate(([0], np.cumsum(chi)))          # bond b's charges at chain_q[chain_off[b] : ...]
    chain_q = np.concatenate(f.charges) if L else np.zeros(0, np.int64)
    bra_bond = np.where(left, site, site + 1)
    out0 = A["out_off"].astype(np.int64)[site] + sec["out_off"].astype(np.int64)
    recs = []
    for p, ra, rb in ((0, np.zeros_like(k0), k0), (1, k0, r1 - r0)):
        sel = np.nonzero(rb > ra)[0]
        al = np.asarray(A["bra_alpha"], np.int64)[bo[sel] + r0[sel] + ra[sel]]
        qb = chain_q[chain_off[bra_bond[sel]] + al]
        q = sec["q"][sel].astype(np.int64)
        r = np.zeros(sel.size, _blk_dt)
        r["site"], r["p"] = site[sel], p
        r["cl"], r["cr"] = np.where(left[sel], qb, q), np.where(left[sel], q, qb)
        r["off"] = out0[sel] + ra[sel] * ncol[sel]
        r["ld"], r["rows"], r["cols"] = ncol[sel], ncol[sel], (rb - ra)[sel]
        r["trans"] = left[sel].astype(np.int32)
        recs.append(r)
    blocks = np.concatenate(recs)
    # the order of the per-site loop: by site, sector, p
    order = np.lexsort((blocks["p"], np.concatenate([np.nonzero(k0 > 0)[0], np.nonzero(r1 - r0 > k0)[0]]), blocks["site"]))
    f.blocks = blocks[order]
    f.dtype = base.dtype
    f.flat = flat_t
    return f


def _fermions_from_slater(mps):
    """Adapter for ``MPSData`` (Slater path, conserve = 'N'): the row-major (merged (p, bra) rows x ket) blocks
    of ``SiteData`` read as column-major matrices without copying."""
    fast = _fermions_from_shard(mps)
    return fast if fast is not None else _fermions_from_sites(mps)


def _fermions_from_sites(mps):
    """The adapter through the per-site objects (assembled, hand-made or sharded ``MPSData``)."""
    if len(mps.sites) != mps.L or any(s_ is None for s_ in mps.sites):
        raise ValueError("the MPS carries no (or only a shard of the) site tensors: convert the whole chain with "
                         "download=True (the default of slater.C_to_MPS)")
    f = _Fermions()
    f.L, f.oc, f.conserve, f.perm = mps.L, mps.ortho_center, "N", None
    f.charges = [np.asarray(b.q_left, np.int64) for b in mps.bonds]
    f.lam_c = np.asarray(mps.bonds[mps.ortho_center].lam)
    recs, arrs = [], []
    f.tabs = [_sector_table(q) for q in f.charges]        # (reused by _Projector.plan: 2 x L calls of numpy.unique less)
    for i, s in enumerate(mps.sites):
        left = s.mode == "left"
        bra_q = f.charges[i] if left else f.charges[i + 1]
        bra_tab = f.tabs[i] if left else f.tabs[i + 1]
        for q, r0, r1, c0, c1, arr in s.blocks:
            c = c1 - c0
            pr = np.asarray(s.bra_p[r0:r1])
            al = np.asarray(s.bra_alpha[r0:r1])
            # rows of one sector: the p = 0 run, then the p = 1 run (stable sort of the merged leg by particle number)
            k0 = int(np.count_nonzero(pr == 0))
            assert not pr[:k0].any() and pr[k0:].all(), "rows of one physical state are contiguous inside a sector"
            for p, ra, rb in ((0, 0, k0), (1, k0, len(pr))):
                if rb == ra:
                    continue
                qb = int(bra_q[al[ra]])
                st, nb = bra_tab[qb]
                assert al[ra] == st and rb - ra == nb, "a block spans whole charge sectors"
                # arr (r x c, row-major) == X (c x r, column-major); the p-run is the column range [ra, rb)
                sub = arr[ra:rb]
                if left:   # rows (p, alpha_L), columns alpha_R: B^p = X[:, ra:rb]^T, cl = qb, cr = q
                    recs.append((i, p, qb, int(q), 0, c, c, rb - ra, 1))
                else:      # rows (p, alpha_R), columns alpha_L: B^p = X[:, ra:rb], cl = q, cr = qb
                    recs.append((i, p, int(q), qb, 0, c, c, rb - ra, 0))
                arrs.append(sub)
    f.blocks = np.array(recs, _blk_dt) if recs else np.zeros(0, _blk_dt)
    f.dtype = np.result_type(*[a.dtype for a in arrs]) if arrs else np.dtype(float)
    flat_t = getattr(mps, "_flat_t", None)
    ok = flat_t is not None and flat_t.numpy().dtype == f.dtype
    if ok:
        base = flat_t.numpy()
        b0, isz, n = base.__array_interface__["data"][0], base.itemsize, base.size
        offs = np.array([a.__array_interface__["data"][0] - b0 for a in arrs], np.int64)
        ok = bool(np.all(offs >= 0) and np.all(offs % isz == 0) and np.all(offs // isz < n)
                  and all(a.flags.c_contiguous for a in arrs))
    if ok:
        f.flat = flat_t
        f.blocks["off"] = offs // isz
    else:      # a hand-made MPSData: pack the blocks
        sizes = np.array([a.size for a in arrs], np.int64)
        o = np.concatenate(([0], np.cumsum((sizes + 1) & ~1)))
        flat = np.zeros(max(int(o[-1]), 2), f.dtype)
        for a, oo in zip(arrs, o[:-1]):
            flat[oo: oo + a.size] = np.ascontiguousarray(a, f.dtype).reshape(-1)
        f.flat = flat
        f.blocks["off"] = o[:-1]
    return f


def _fermions_from_dense(T, q, lam_c, oc, conserve):
    """Adapter for dense site tensors T[i] (2, chi_l, chi_r) with per-bond charge arrays q[b] (any order):
    indices are sorted by charge (stable) and the non-zero charge blocks packed.  Used for ``PfMPSData``
    (conserve = 'parity') and by the tests."""
    f = _Fermions()
    f.L, f.oc, f.conserve = len(T), oc, conserve
    perm = [np.argsort(np.asarray(x), kind="stable") for x in q]
    f.charges = [np.asarray(x, np.int64)[pm] for x, pm in zip(q, perm)]
    f.lam_c = np.asarray(lam_c)[perm[oc]]
    f.dtype = np.result_type(*[t.dtype for t in T], float)
    recs, arrs, o = [], [], 0
    for i, t in enumerate(T):
        tl, tr = _sector_table(f.charges[i]), _sector_table(f.charges[i + 1])
        t = np.asarray(t)[:, perm[i]][:, :, perm[i + 1]]
        for p in (0, 1):
            for cl, (a, n) in tl.items():
                for cr, (b, m) in tr.items():
                    blk = t[p, a:a + n, b:b + m]
                    if not np.any(blk):
                        continue
                    recs.append((i, p, cl, cr, o, n, n, m, 0))
                    arrs.append(np.asfortranarray(blk))
                    o += (n * m + 1) & ~1
    flat = np.zeros(max(o, 2), f.dtype)
    for r, a in zip(recs, arrs):
        flat[r[4]: r[4] + a.size] = a.reshape(-1, order="F")
    f.flat, f.blocks = flat, (np.array(recs, _blk_dt) if recs else np.zeros(0, _blk_dt))
    f.perm = perm            # index order of every bond after the sort by charge
    return f


def _fermions_from_imps(imps):
    """Adapter for the unit cell of an infinite MPS (``iMPS.iMPSData``, all tensors in B form).  The labels of the last
    bond are those of bond 0 plus the charge of one cell, so that q_l + p = q_r holds on every site
    (TeNPy: ``mps.gauge_total_charge(qtotal=mps.get_total_charge())``, gutzwiller.py:202-212 / :392-398)."""
    kind = getattr(imps, "conserve", None)
    if kind not in ("N", "parity"):
        raise ValueError(f"FermionSite must conserve either 'N' or 'parity', found {kind!r}")        # gutzwiller.py:172-176
    if any(f_ != "B" for f_ in imps.form):
        raise ValueError("the infinite MPS is not in right-canonical form")
    q = [np.asarray(x, np.int64) for x in imps.charges[: imps.L]]
    last = np.asarray(imps.charges[0], np.int64) + int(imps.cell_charge)
    q.append(last % 2 if kind == "parity" else last)
    if kind == "parity":
        q = [x % 2 for x in q]
    f = _fermions_from_dense(imps.dense_tensors(), q, np.ones(len(q[imps.L])), imps.L, kind)
    f.cell_charge = int(imps.cell_charge)
    f.infinite = True
    return f


def infer_parities(T, tol=1e-9):
    """Fermion parity to the left of every index of every bond of a parity-conserving MPS given as dense tensors
    T[i] (2, chi_l, chi_r): parity(right index) = parity(left index) + p (mod 2), read off the dominant entry of
    each column and verified on all entries above ``tol`` of the site's largest one."""
    q = [np.zeros(T[0].shape[1], np.int64)]
    for i, t in enumerate(T):
        a = np.abs(np.asarray(t))
        src = (q[-1][None, :, None] + np.arange(2)[:, None, None]) % 2 + 0 * a.astype(np.int64)     # parity a column would get
        flat = a.reshape(-1, a.shape[2])
        nxt = src.reshape(-1, a.shape[2])[flat.argmax(axis=0), np.arange(a.shape[2])]
        bad = (a > tol * max(a.max(), 1e-300)) & (src != nxt[None, None, :])
        if bad.any():
            raise ValueError(f"site {i} does not conserve the fermion parity")
        q.append(nxt.astype(np.int64))
    return q


def native(mps):
    """The package's own container behind a TeNPy ``MPS`` it returned (``to_tenpy`` leaves a reference on the object), or
    ``mps`` itself.  A TeNPy ``MPS`` from elsewhere, or one modified after the conversion, is not supported as input."""
    own = getattr(mps, "_temfpy_amd", None)
    if own is not None:
        dims = [len(x) for x in own.lam]          # TeNPy's chi: bonds 1 .. L-1 of a finite MPS, bonds 0 .. L-1 of an infinite one
        if getattr(mps, "L", own.L) != own.L or list(getattr(mps, "chi", [])) not in ([], dims[1:-1], dims[:-1]):
            raise ValueError("the TeNPy MPS was modified after the conversion; convert with as_tenpy=False and pass that result")
        return own
    return mps


def _as_fermions(mps):
    from .mps_data import MPSData

    from .iMPS import iMPSData

    if isinstance(mps, _Fermions):
        return mps
    mps = native(mps)
    if isinstance(mps, iMPSData):
        return _fermions_from_imps(mps)
    if isinstance(mps, MPSData):
        return _fermions_from_slater(mps)
    if isinstance(mps, SpinMPSData):      # (a projected spin chain as input of iMPS.MPS_to_iMPS: 2 S^z blocks or one trivial sector)
        if mps.form != ["B"] * mps.L:
            raise ValueError("the spin MPS is not in canonical form (return_canonical=False)")
        return _fermions_from_dense(mps.dense_tensors(), mps.charges, mps.lam[0], 0, "Sz" if mps.conserve == "Sz" else "none")
    if hasattr(mps, "bonds") and hasattr(mps.bonds[0], "parity"):      # PfMPSData
        T = mps.dense_tensors()
        return _fermions_from_dense(T, infer_parities(T), mps.bonds[mps.ortho_center].lam, mps.ortho_center, "parity")
    raise TypeError(f"expected the MPS returned by slater.C_to_MPS / pfaffian.C_to_MPS, got {type(mps)!r}")


# ---------------------------------------------------------------------------------------------------
# device pipeline
# ---------------------------------------------------------------------------------------------------
def _cdiv(a, b):
    return -(-a // b)


class _Launches:
    """Descriptor tables of many launches of one kind, uploaded together."""

    def __init__(self, dtype):
        self.descs, self.spans, self.n, self.dtype = [], [], 0, dtype

    def add(self, recs):
        """recs: structured array of one launch; returns the launch id."""
        self.spans.append((self.n, len(recs)))
        self.descs.append(recs)
        self.n += len(recs)
        return len(self.spans) - 1

    def table(self):
        if not self.descs:
            return np.zeros(0, self.dtype)
        # (as bytes: np.concatenate of structured arrays promotes the field list once per array, 20 ms for a thousand launches)
        sz = self.dtype.itemsize
        raw = np.concatenate([np.ascontiguousarray(d_).view(np.uint8).reshape(len(d_), sz) for d_ in self.descs])
        return raw.reshape(-1).view(self.dtype)


class _Arena:
    def __init__(self, elem):
        self.n, self.elem = 0, elem

    def take(self, count):
        o = self.n
        self.n += (int(count) + 1) & ~1        # 16-byte granularity
        return o


def _gemm_recs(items):
    """items: list of (A, B, C, M, N, K, lda, ldb, ldc) -> (gemm_desc records, tiles, tile_n) of one launch."""
    d = np.zeros(len(items), nat.gemm_desc)
    if items:
        a = np.array(items, np.int64)
        for k, f in enumerate(("A", "B", "C", "M", "N", "K", "lda", "ldb", "ldc")):
            d[f] = a[:, k]
        d["lda"], d["ldb"], d["ldc"] = np.maximum(d["lda"], 1), np.maximum(d["ldb"], 1), np.maximum(d["ldc"], 1)
    return d


def _gemm_tiles(d):
    tn = 16 if (len(d) and int(d["N"].max()) <= 16) else 64
    tm = _cdiv(d["M"].astype(np.int64), 64)
    tnn = _cdiv(d["N"].astype(np.int64), tn)
    cnt = tm * tnn
    total = int(cnt.sum())
    prob = np.repeat(np.arange(len(d)), cnt)
    local = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    tiles = np.zeros((total, 4), np.int32)
    tiles[:, 0], tiles[:, 1], tiles[:, 2] = prob, local % tm[prob], local // tm[prob]
    return tiles, tn


def _gemm_tiles_spans(d, spans):
    """The tile tables of every launch of a GEMM descriptor table at once: (tiles, [(first tile, count)], [tile_n])."""
    if not spans:
        return np.zeros((0, 4), np.int32), [], []
    starts = np.array([o for o, _ in spans], np.int64)
    lens = np.array([n for _, n in spans], np.int64)
    N, M = d["N"].astype(np.int64), d["M"].astype(np.int64)
    tn_span = np.full(len(spans), 64, np.int64)
    nz = lens > 0
    if nz.any():
        mx = np.maximum.reduceat(N, starts[nz]) if len(N) else np.zeros(0, np.int64)
        # (reduceat runs to the next start: spans are contiguous and in order, empty ones skipped)
        tn_span[nz] = np.where(mx <= 16, 16, 64)
    span_of = np.repeat(np.arange(len(spans)), lens)
    tn = tn_span[span_of]
    tm = _cdiv(M, 64)
    cnt = tm * _cdiv(N, tn)
    total = int(cnt.sum())
    tiles = np.zeros((total, 4), np.int32)
    tiles[:, 0] = np.repeat((np.arange(len(d)) - starts[span_of]).astype(np.int32), cnt)      # problem index inside its launch
    local = np.arange(total, dtype=np.int32) - np.repeat((np.cumsum(cnt) - cnt).astype(np.int32), cnt)
    tiles[:, 2], tiles[:, 1] = np.divmod(local, np.repeat(tm.astype(np.int32), cnt))
    per_span = np.bincount(span_of, weights=cnt, minlength=len(spans)).astype(np.int64)
    first = np.cumsum(per_span) - per_span
    return tiles, [(int(a), int(b)) for a, b in zip(first, per_span)], [int(x) for x in tn_span]


def _dominant_eigenpair(apply, v0, tol=2e-15, krylov=24, restarts=400, stagnation=1e-11):
    """Eigenvalue of largest modulus and its eigenvector of the linear map `apply` (host vector -> host vector) by restarted
    Arnoldi iteration: the role of scipy's ARPACK behind TeNPy's ``TransferMatrix.eigenvectors``.  Only the Krylov
    bookkeeping (dot products over <= `krylov` vectors, one small Hessenberg eigenproblem per restart) is done here."""
    v = np.asarray(v0)
    v = v / np.linalg.norm(v)
    m = min(krylov, v.size)
    last_res = np.inf
    for _ in range(restarts):
        V = np.zeros((m + 1, v.size), v.dtype)
        H = np.zeros((m + 1, m), v.dtype)
        V[0], used, beta = v, m, 0.0
        for k in range(m):
            w = apply(V[k])
            size = float(np.linalg.norm(w))
            for _pass in range(2):                     # classical Gram-Schmidt, twice
                h = V[: k + 1].conj() @ w
                w = w - h @ V[: k + 1]
                H[: k + 1, k] += h
            beta = float(np.linalg.norm(w))
            if beta <= 1e-11 * size or k + 1 == v.size:
                # the Krylov space is exhausted (an invariant subspace: the vectors may live in fewer dimensions than their
                # length says); what is left of w is rounding noise and must not become a basis vector
                used, beta = k + 1, 0.0
                break
            H[k + 1, k] = beta
            V[k + 1] = w / beta
        ev, Yv = np.linalg.eig(H[:used, :used])
        i = int(np.argmax(np.abs(ev)))
        theta, y = ev[i], Yv[:, i]
        res = abs(beta * y[-1]) / max(np.linalg.norm(y), 1e-300)     # Ritz residual (of the complex vector, if it is one)
        if v.dtype.kind != "c":
            # A real map: a real leading Ritz value has a real vector.  A complex PAIR in front has no real eigenvector; its
            # value is returned as it is (callers demand a real positive dominant eigenvalue and raise), and the restart
            # vector is the real part, which together with its image spans the plane of the pair.
            if not abs(theta.imag) > 1e-9 * abs(theta):
                theta = theta.real
            y = y.real
        v = y @ V[:used]
        v = v / np.linalg.norm(v)
        # done at `tol`, or when the residual has stopped falling at the rounding level of `apply` (the Schmidt values of
        # the boundary bond are square roots of Gram eigenvalues: an eigenvector good to 1e-13 leaves noise of 3e-7 there)
        if res <= tol * abs(theta) or (res <= stagnation * abs(theta) and res > 0.5 * last_res):
            return complex(theta), v
        last_res = res
    raise np.linalg.LinAlgError("Arnoldi iteration for the dominant eigenvector of the transfer matrix did not converge")


class _Projector:
    """Builds and runs the device pipeline for one projection (see the module docstring)."""

    def __init__(self, fer, pairs, keep_fn, cutoff, device, method="parallel", shift=None):
        import torch

        if not torch.cuda.is_available():
            raise nat.NativeError("temfpy_amd.gutzwiller needs a HIP device; there is no CPU fallback")
        self.torch, self.device = torch, torch.device(device)
        self.lib = nat.load()
        self.f, self.pairs, self.keep_fn, self.cutoff = fer, pairs, keep_fn, cutoff
        if method not in ("sequential", "parallel"):
            raise ValueError(f"`method` must be 'sequential' or 'parallel', got {method!r}")
        self.method = method
        # infinite MPS: `shift` maps a sector label of the last bond of the unit cell to the label of the same sector on
        # bond 0 (the charge of one cell subtracted); the cell is then closed by the fixed points of its transfer matrix
        self.shift = shift
        if shift is not None:
            if method != "parallel":
                raise ValueError("infinite MPS: only method='parallel' is available")
            self.cutoff = max(cutoff, _INF_SCHMIDT_EPS)
        self.cplx = np.dtype(fer.dtype).kind == "c"
        self.dt = nat.TMF_C128 if self.cplx else nat.TMF_F64
        self.np_dt = np.dtype(np.complex128 if self.cplx else np.float64)
        self.elem = self.np_dt.itemsize
        self.timings = {}

    # ---- structure ------------------------------------------------------------------------------
    def plan(self):
        f = self.f
        Ls = f.L // 2
        tabs = getattr(f, "tabs", None) or [_sector_table(q) for q in f.charges]
        fb = f.blocks                 # (column lists: iterating the records of a structured array costs 1 us per field)
        keys = list(zip(fb["site"].tolist(), fb["p"].tolist(), fb["cl"].tolist()))
        blk = dict(zip(keys, zip(fb["cr"].tolist(), range(len(fb)))))
        assert len(blk) == len(keys), "a physical state maps a left charge sector to one right sector"
        kept = []
        for j in range(Ls + 1):
            kept.append({c: n for c, (st, n) in tabs[2 * j].items() if self.keep_fn(j, c)})
        # spin blocks (j, sigma, c -> c') with their two fermion factors
        sb = []
        for j in range(Ls):
            for sg, (p1, p2) in enumerate(self.pairs):
                for c in kept[j]:
                    a = blk.get((2 * j, p1, c))
                    if a is None:
                        continue
                    b = blk.get((2 * j + 1, p2, a[0]))
                    if b is None or b[0] not in kept[j + 1]:
                        continue
                    sb.append((j, sg, c, b[0], a[1], b[1], a[0]))
        # prune sectors that are not connected to both ends of the chain
        byj = [[] for _ in range(Ls)]
        for x in sb:
            byj[x[0]].append(x)
        if self.shift is None:
            alive = [set() for _ in range(Ls + 1)]
            alive[0] = set(kept[0])
            for j in range(Ls):
                alive[j + 1] = {x[3] for x in byj[j] if x[2] in alive[j]}
            back = [set() for _ in range(Ls + 1)]
            back[Ls] = alive[Ls]
            for j in range(Ls - 1, -1, -1):
                back[j] = {x[2] for x in byj[j] if x[3] in back[j + 1]} & alive[j]
        else:      # the cell repeats: a sector survives if it lies on a closed path through the cell
            alive = [set(k) for k in kept]
            while True:
                before = sum(len(a) for a in alive)
                alive[0] &= {self.shift(c) for c in alive[Ls]}
                alive[Ls] = {c for c in alive[Ls] if self.shift(c) in alive[0]}
                for j in range(Ls):
                    alive[j + 1] &= {x[3] for x in byj[j] if x[2] in alive[j]}
                for j in range(Ls - 1, -1, -1):
                    alive[j] &= {x[2] for x in byj[j] if x[3] in alive[j + 1]}
                if sum(len(a) for a in alive) == before:
                    break
            back = alive
            for c in alive[Ls]:
                assert kept[Ls][c] == kept[0][self.shift(c)], "the two ends of the unit cell carry the same bond"
        self.sb = [[x for x in byj[j] if x[2] in back[j] and x[3] in back[j + 1]] for j in range(Ls)]
        self.sect = [{c: kept[j][c] for c in sorted(back[j])} for j in range(Ls + 1)]   # bond -> {c: n}
        self.tabs, self.Ls = tabs, Ls
        if any(len(s) == 0 for s in self.sect):
            raise ValueError("the projected state vanishes: no charge sector connects the two ends of the chain")
        if self.shift is not None:
            self.end_of = {self.shift(c): c for c in self.sect[Ls]}      # bond-0 label -> label on the last bond
            self.sect[Ls] = {self.end_of[q]: n for q, n in self.sect[0].items()}      # same index order on both copies

    # ---- run ------------------------------------------------------------------------------------
    def run(self, canonical=True):
        """(The collector is paused for the call: the tables are tens of thousands of small Python objects, and a collection
        of the oldest generation in the middle of building them cost 40 - 50 ms in one call out of three.)"""
        import gc

        was = gc.isenabled()
        gc.disable()
        try:
            return self._run(canonical)
        finally:
            if was:
                gc.enable()

    def _run(self, canonical=True):
        torch, lib, f = self.torch, self.lib, self.f
        t0 = time.perf_counter()
        self.plan()
        self.timings["setup: plan"] = time.perf_counter() - t0
        Ls, el = self.Ls, self.elem
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ar = _Arena(el)
        # -- fermion blocks F (regrouped, column-major) and centre scaling matrices
        need = sorted({x[4] for row in self.sb for x in row} | {x[5] for row in self.sb for x in row})
        Foff = {}
        b_rows, b_cols, b_trans, b_site = (f.blocks[x].tolist() for x in ("rows", "cols", "trans", "site"))
        for k in need:
            n_l, n_r = (b_cols[k], b_rows[k]) if b_trans[k] else (b_rows[k], b_cols[k])
            Foff[k] = (ar.take(n_l * n_r), n_l, n_r)
        scale_blocks = [k for k in need if b_site[k] == f.oc and f.oc < f.L]
        Goff = {k: ar.take(Foff[k][1] * Foff[k][2]) for k in scale_blocks}
        # -- spin blocks T, V (left-merged), V2, W (right-merged), Y, Z, Vz, Bh, X, R
        Toff = {}
        for j in range(Ls):
            for x in self.sb[j]:
                Toff[(j, x[1], x[2])] = ar.take(self.sect[j][x[2]] * self.sect[j + 1][x[3]])
        Vinfo, Winfo = [], []          # per site: {c': (off, m, {(sg, c): row0})}, {c: (off, w, {(sg, c'): col0})}
        for j in range(Ls):
            vi, wi = {}, {}
            for cp in self.sect[j + 1]:
                rows, m = {}, 0
                for x in sorted((x for x in self.sb[j] if x[3] == cp), key=lambda x: (x[1], x[2])):
                    rows[(x[1], x[2])] = m
                    m += self.sect[j][x[2]]
                vi[cp] = [ar.take(m * self.sect[j + 1][cp]), m, rows]
            for c in self.sect[j]:
                cols, w = {}, 0
                for x in sorted((x for x in self.sb[j] if x[2] == c), key=lambda x: (x[1], x[3])):
                    cols[(x[1], x[3])] = w
                    w += self.sect[j + 1][x[3]]
                wi[c] = [None, w, cols]
            Vinfo.append(vi)
            Winfo.append(wi)
        mx_w = max(sum(self.sect[j][c] * v[1] + 2 for c, v in Winfo[j].items()) for j in range(Ls))
        Wo = ar.take(mx_w)                                  # transient W of the leftward sweep
        # per bond and sector: R (from the rightward sweep), L (leftward sweep), C = R L / norm, its right singular
        # vectors Vz and a workspace; per site and left sector: Q~ = B^H (w x n), Q~ Vz, final B^H
        sq = lambda: [{c: ar.take(n * n) for c, n in s.items()} for s in self.sect]     # noqa: E731
        Rb, Lb, Cb, Vzb, Jwb = sq(), sq(), sq(), sq(), sq()
        Yq, G1o, Bho, So = [], [], [], []
        n_sv = 0
        for j in range(Ls):
            Yq.append({c: ar.take(v[1] * self.sect[j][c]) for c, v in Winfo[j].items()})
            G1o.append({c: ar.take(v[1] * self.sect[j][c]) for c, v in Winfo[j].items()})
        for j in range(Ls):
            Bho.append({c: ar.take(v[1] * self.sect[j][c]) for c, v in Winfo[j].items()})
        for j in range(Ls + 1):
            so = {}
            for c, n in self.sect[j].items():
                so[c] = n_sv
                n_sv += n
            So.append(so)
        cnt_index = {}
        for j in range(Ls + 1):
            for c in self.sect[j]:
                cnt_index[(j, c)] = len(cnt_index)
        n_sec_tot = len(cnt_index)
        for b_, tab in ((0, Rb), (Ls, Lb)):
            if self.shift is None and (len(self.sect[b_]) != 1 or next(iter(self.sect[b_].values())) != 1):
                raise ValueError("the ends of the chain must carry a single state")
        if self.shift is not None:     # second copies of the first and last tensor of the cell (boundary gauge, _close_cell)
            Tping = {k: ar.take(self.sect[k[0]][k[2]] * self.sect[k[0] + 1][x3]) for k, x3 in
                     (((x[0], x[1], x[2]), x[3]) for x in self.sb[Ls - 1])}
            Tpong = {k: ar.take(self.sect[k[0]][k[2]] * self.sect[k[0] + 1][x3]) for k, x3 in
                     (((x[0], x[1], x[2]), x[3]) for x in self.sb[0])}

        self.timings["setup: layout"] = time.perf_counter() - t0 - self.timings["setup: plan"]
        t_al = time.perf_counter()
        d_ar = torch.zeros(ar.n, dtype=torch.complex128 if self.cplx else torch.float64, device=self.device)
        base = d_ar.data_ptr()
        P = lambda off: base + el * off                      # noqa: E731
        d_sv = torch.zeros(max(n_sv, 1), dtype=torch.float64, device=self.device)
        d_cnt = torch.zeros(max(n_sec_tot, 1), dtype=torch.int32, device=self.device)

        # -- upload of the fermion tensors (one copy) and regrouping
        if isinstance(f.flat, np.ndarray):
            d_flat = torch.from_numpy(f.flat).to(self.device)
        else:
            d_flat = f.flat.to(self.device, non_blocking=True)
        fbase = d_flat.data_ptr()
        cp = np.zeros(len(need), nat.copy_desc)
        if need:
            sel = f.blocks[np.asarray(need, np.int64)]
            cp["src"], cp["dst"] = fbase + el * sel["off"].astype(np.int64), base + el * np.array([Foff[k][0] for k in need], np.int64)
            cp["rows"], cp["cols"], cp["lds_"] = sel["rows"], sel["cols"], sel["ld"]
            cp["ldd"], cp["flags"] = np.array([Foff[k][1] for k in need], np.int64), np.where(sel["trans"] != 0, 1, 0)
        keep_alive = [d_flat]
        self.timings["setup: arena + upload"] = time.perf_counter() - t_al
        self._copy(cp, stream, keep_alive)
        Fptr = {k: P(v[0]) for k, v in Foff.items()}
        if scale_blocks:      # Lambda on the left index of the centre tensor: F' = diag(lam) F by the GEMM kernel
            tab = self.tabs[f.oc]
            host_d = np.zeros(sum(Foff[k][1] ** 2 for k in scale_blocks) + 2, self.np_dt)
            items, o = [], 0
            done = {}
            for k in scale_blocks:
                cl = int(f.blocks[k]["cl"])
                n_l, n_r = Foff[k][1], Foff[k][2]
                if cl not in done:
                    st, n = tab[cl]
                    host_d[o: o + n * n] = np.diag(f.lam_c[st: st + n]).astype(self.np_dt).reshape(-1)
                    done[cl] = o
                    o += n * n
                items.append((0, Fptr[k], P(Goff[k]), n_l, n_r, n_l, n_l, n_l, n_l, done[cl]))
            d_D = torch.from_numpy(host_d).to(self.device)
            keep_alive.append(d_D)
            it2 = [(d_D.data_ptr() + el * it[9],) + it[1:9] for it in items]
            self._gemm_now(it2, 0, stream, keep_alive)
            for k in scale_blocks:
                Fptr[k] = P(Goff[k])
        # -- pair products: ONE batched launch
        items = []
        for j in range(Ls):
            for x in self.sb[j]:
                n, nm, npr = self.sect[j][x[2]], Foff[x[4]][2], self.sect[j + 1][x[3]]
                items.append((Fptr[x[4]], Fptr[x[5]], P(Toff[(j, x[1], x[2])]), n, npr, nm, n, nm, n))
        self._gemm_now(items, 0, stream, keep_alive)
        self.timings["setup+pairs"] = time.perf_counter() - t0
        if not canonical:
            torch.cuda.synchronize(self.device)
            h = d_ar.cpu().numpy()
            blocks = []
            off = [{c: o for c, o in zip(s, np.concatenate(([0], np.cumsum(list(s.values()))))[:-1])} for s in self.sect]
            for j in range(Ls):
                bl = []
                for x in self.sb[j]:
                    n, npr = self.sect[j][x[2]], self.sect[j + 1][x[3]]
                    o = Toff[(j, x[1], x[2])]
                    a = h[o: o + n * npr].reshape(npr, n).T
                    bl.append((x[1], x[2], x[3], int(off[j][x[2]]), int(off[j][x[2]]) + n, int(off[j + 1][x[3]]),
                               int(off[j + 1][x[3]]) + npr, a))
                blocks.append(bl)
            dims = [sum(s.values()) for s in self.sect]
            lam = [np.ones(d) / np.sqrt(d) for d in dims]       # gutzwiller.py:258 / :460
            ch = [np.concatenate([[c] * n for c, n in s.items()]) for s in self.sect]
            return blocks, lam, ch, None

        # ================= descriptor tables =================
        t1 = time.perf_counter()
        if self.shift is None:
            d_ar[Rb[0][next(iter(self.sect[0]))]] = 1.0           # R of the (empty) left end, L of the right end
            d_ar[Lb[Ls][next(iter(self.sect[Ls]))]] = 1.0
            cell = None
        else:
            from types import SimpleNamespace

            cell = self._close_cell(SimpleNamespace(P=P, d_ar=d_ar, Toff=Toff, Rb=Rb, Lb=Lb, Vinfo=Vinfo, Winfo=Winfo, Yq=Yq, Wo=Wo,
                                                    stream=stream, keep=keep_alive, Tping=Tping, Tpong=Tpong))
            self.timings["fixed points"] = time.perf_counter() - t1
            self.timings["cell applications"] = self.cell_applications
        # Power-of-two rescaling of the triangular factors after every step (TeNPy's canonical_form_finite renormalises every
        # step): the norm of the projected state falls by a constant factor per site and would leave the range of a double
        # after ~1300 spins.  exps[0 / 1]: running exponents of the two sweeps; exps[2 + j] / exps[2 + Ls + 1 + j]: their values
        # when R_j / L_j were written.  (Infinite cells are short: not rescaled.)
        d_exp = torch.zeros(2 + 2 * (Ls + 1), dtype=torch.int64, device=self.device)
        keep_alive.append(d_exp)
        rescaling = self.shift is None and os.environ.get("TMF_GW_RESCALE", "1") != "0"

        def make(sites1, sites2, with_tail):
            """Descriptor tables of the sweep steps of the given sites (and of the all-bond tail), uploaded; the launch closures."""
            from types import SimpleNamespace as _NS

            t_mk = time.perf_counter()
            G, CP, QR, RS = _Launches(nat.gemm_desc), _Launches(nat.copy_desc), _Launches(nat.qr_desc), _Launches(nat.rescale_desc)
            steps1, steps2 = [], []
            gc = gc_bond = None
            # rightward sweep: V_j = R_j T_j (left-merged), QR in place, R -> bond j+1
            for j in sites1:
                g = []
                for x in self.sb[j]:
                    _, sg, c, cp_, *_ = x
                    voff, m, rows = Vinfo[j][cp_]
                    n, npr = self.sect[j][c], self.sect[j + 1][cp_]
                    g.append((P(Rb[j][c]), P(Toff[(j, sg, c)]), P(voff + rows[(sg, c)]), n, npr, n, n, n, m))
                qd = np.zeros(len(Vinfo[j]), nat.qr_desc)
                rs = np.zeros(len(Vinfo[j]), nat.rescale_desc)
                for i, (cp_, (voff, m, rows)) in enumerate(Vinfo[j].items()):
                    npr = self.sect[j + 1][cp_]
                    qd[i] = (P(voff), P(Rb[j + 1][cp_]), m, npr, m, npr, 2 if self.method == "parallel" else 4, 0)   # parallel: nothing reads Q; else: every Q after the sweep
                    rs[i] = (P(Rb[j + 1][cp_]), npr, npr, npr, 0)
                steps1.append((G.add(_gemm_recs(g)), QR.add(qd), RS.add(rs)))
            sv_ptr, cnt_ptr = d_sv.data_ptr(), d_cnt.data_ptr()
            jd, tail = np.zeros(n_sec_tot if with_tail else 0, nat.jacobi_desc), None
            for j in range(Ls + 1 if with_tail else 0):
                for c, n in self.sect[j].items():
                    i = cnt_index[(j, c)]
                    # Jacobi without accumulator on the conjugate transpose of the QR-preconditioned factor (graded
                    # columns): its sorted, normalised left singular vectors are the wanted right ones
                    jd[i] = (P(Jwb[j][c]), 0, P(Vzb[j][c]), sv_ptr + 8 * So[j][c], cnt_ptr + 4 * i, self.cutoff ** 2, n, n, n, n)
            if self.method == "sequential":
                # leftward sweep, one SVD per site as TeNPy does it (X = U S of bond j+1 lives in the L buffers):
                # N = A_j X_{j+1}, Y = N^H = Q R, second QR R^H = Q3 R3 (preconditioner: R3 is nearly diagonal), Jacobi on
                # R3^H gives the right singular vectors V3 of R3, so N = (Q3 U3) S (Q V3)^H: B^H = Q V3, X_j = N B^H
                for j in range(Ls - 1, -1, -1):
                    g, cpy, gb, gx, wl, o = [], [], [], [], {}, 0
                    for c, v in Winfo[j].items():
                        wl[c] = o
                        o += (self.sect[j][c] * v[1] + 1) & ~1
                    for x in self.sb[j]:
                        _, sg, c, cp_, *_ = x
                        voff, m, rows = Vinfo[j][cp_]
                        n, npr = self.sect[j][c], self.sect[j + 1][cp_]
                        cols = Winfo[j][c][2]
                        g.append((P(voff + rows[(sg, c)]), P(Lb[j + 1][cp_]), P(Wo + wl[c] + n * cols[(sg, cp_)]),
                                  n, npr, npr, m, npr, n))
                    qd, qd2 = np.zeros(len(Winfo[j]), nat.qr_desc), np.zeros(len(Winfo[j]), nat.qr_desc)
                    for i, (c, v) in enumerate(Winfo[j].items()):
                        n, w = self.sect[j][c], v[1]
                        cpy.append((P(Wo + wl[c]), P(Yq[j][c]), n, w, n, w, 3 if self.cplx else 1, 0))
                        qd[i] = (P(Yq[j][c]), P(Cb[j][c]), w, n, w, n, 1, 0)
                        qd2[i] = (P(Cb[j][c]), P(Jwb[j][c]), n, n, n, n, 3, 0)       # R3^H only, Q3 is not needed
                        gb.append((P(Yq[j][c]), P(Vzb[j][c]), P(Bho[j][c]), w, n, n, w, n, w))
                        gx.append((P(Wo + wl[c]), P(Bho[j][c]), P(Lb[j][c]), n, n, w, n, w, n))
                    i0 = cnt_index[(j, next(iter(self.sect[j])))]
                    steps2.append((G.add(_gemm_recs(g)), CP.add(np.array(cpy, nat.copy_desc)), QR.add(qd), QR.add(qd2),
                                   (i0, len(self.sect[j]), max(self.sect[j].values())), G.add(_gemm_recs(gb)),
                                   G.add(_gemm_recs(gx))))
            else:
                # leftward sweep: W_j = T_j L_{j+1} (right-merged), Y = W^H = Q~ R, L_j = R^H
                for j in sites2:
                    g, cpy, wl, o = [], [], {}, 0
                    for c, v in Winfo[j].items():
                        wl[c] = o
                        o += (self.sect[j][c] * v[1] + 1) & ~1
                    for x in self.sb[j]:
                        _, sg, c, cp_, *_ = x
                        n, npr = self.sect[j][c], self.sect[j + 1][cp_]
                        cols = Winfo[j][c][2]
                        g.append((P(Toff[(j, sg, c)]), P(Lb[j + 1][cp_]), P(Wo + wl[c] + n * cols[(sg, cp_)]), n, npr, npr, n, npr, n))
                    qd = np.zeros(len(Winfo[j]), nat.qr_desc)
                    rs = np.zeros(len(Winfo[j]), nat.rescale_desc)
                    for i, (c, v) in enumerate(Winfo[j].items()):
                        n, w = self.sect[j][c], v[1]
                        cpy.append((P(Wo + wl[c]), P(Yq[j][c]), n, w, n, w, 3 if self.cplx else 1, 0))
                        qd[i] = (P(Yq[j][c]), P(Lb[j][c]), w, n, w, n, 1 | 4, 0)      # 4: Q~ of every site in one launch after the sweep
                        rs[i] = (P(Lb[j][c]), n, n, n, 0)
                    steps2.append((G.add(_gemm_recs(g)), CP.add(np.array(cpy, nat.copy_desc)), QR.add(qd), RS.add(rs)))
                gc = gc_bond = None
                if with_tail:
                    # batched tail: C = R L / norm, SVD of every bond and sector at once, B^H = blockdiag(Vz_{j+1})^H (Q~ Vz_j)
                    gc, g1, g2, gc_bond = [], [], [], []
                    for j in range(Ls + 1):
                        for c, n in self.sect[j].items():
                            gc.append((P(Rb[j][c]), P(Lb[j][c]), P(Cb[j][c]), n, n, n, n, n, n))
                            gc_bond.append(j)
                    for j in range(Ls):
                        for c, v in Winfo[j].items():
                            n, w = self.sect[j][c], v[1]
                            g1.append((P(Yq[j][c]), P(Vzb[j][c]), P(G1o[j][c]), w, n, n, w, n, w))
                            for (sg, cp_), c0 in v[2].items():
                                npr = self.sect[j + 1][cp_]
                                g2.append((P(Vzb[j + 1][cp_]), P(G1o[j][c] + c0), P(Bho[j][c] + c0), npr, n, npr, npr, w, w))
                    qc = np.zeros(n_sec_tot, nat.qr_desc)
                    for j in range(Ls + 1):
                        for c, n in self.sect[j].items():
                            qc[cnt_index[(j, c)]] = (P(Cb[j][c]), P(Jwb[j][c]), n, n, n, n, 3, 0)      # C = Q R, R^H -> Jacobi
                    qc = qc[np.argsort(-qc["n"], kind="stable")]
                    tail = (G.add(_gemm_recs(gc)), G.add(_gemm_recs(g1)), G.add(_gemm_recs(g2)), QR.add(qc))
                    jd = jd[np.argsort(-jd["p"], kind="stable")]          # one launch over all bonds: large problems first
            # upload all tables
            gt = G.table()
            tiles, tile_span, tile_n = _gemm_tiles_spans(gt, G.spans)
            tabs_h = {"g": gt, "t": tiles, "cp": CP.table(), "qr": QR.table(), "jc": jd, "rs": RS.table()}
            qr_max = [(int(tabs_h["qr"][o: o + n]["m"].max()), int(tabs_h["qr"][o: o + n]["n"].max())) for o, n in QR.spans]
            # The same factorisations through the on-chip slab kernel (panel columns in registers, reflector blocks in
            # LDS: 2x faster than the L2-resident kernel at 280 x 140) whenever the rows fit its registers; Q is built
            # in a scratch (one per sweep direction) and copied over the block.
            slab_rows = 1024 if self.cplx else 2048
            use_slab = os.environ.get("TMF_GW_QR", "slab") == "slab" and max(q[0] for q in qr_max) <= slab_rows
            later = (tabs_h["qr"]["flags"] & 4) != 0            # factorisations whose Q is formed after the sweep
            if not use_slab or os.environ.get("TMF_GW_Q_LATER", "1") == "0":
                tabs_h["qr"]["flags"] &= ~np.int32(4)
                later[:] = False
            tabs_d = {k: torch.from_numpy(v.view(np.uint8).reshape(-1).copy() if v.size else np.zeros(16, np.uint8)).to(self.device)
                      for k, v in tabs_h.items()}
            keep_alive.append(tabs_d)
            formq = None
            if use_slab:
                qt = tabs_h["qr"]
                sl = np.zeros(len(qt), nat.slab_desc)
                sl["A"], sl["R"], sl["n"], sl["c"], sl["lda"], sl["ldr"] = qt["A"], qt["R"], qt["m"], qt["n"], qt["lda"], qt["ldr"]
                sl["flags"] = (qt["flags"] & 1) | np.where(qt["flags"] & 2, 4, 0) | np.where(later, 8, 0)
                sl["ldq"] = qt["m"]
                sizes = (qt["m"].astype(np.int64) * qt["n"] + 1) & ~1
                need, rel = 0, np.zeros(len(qt), np.int64)
                for o, n in QR.spans:
                    c_ = np.concatenate(([0], np.cumsum(sizes[o: o + n])))
                    rel[o: o + n] = c_[:-1]
                    need = max(need, int(c_[-1]))
                scratch = [torch.empty(need + 2, dtype=d_ar.dtype, device=self.device) for _ in range(2)]
                keep_alive.append(scratch)
                first2 = min([x[2] for x in steps2] + ([tail[3]] if tail is not None else []), default=len(QR.spans))
                for i, (o, n) in enumerate(QR.spans):      # launches of the rightward sweep come first in the table
                    sl["Q"][o: o + n] = scratch[0 if i < first2 else 1].data_ptr() + el * rel[o: o + n]
                if later.any():      # their reflector scalars: one buffer for all; the launch that forms every Q, large slabs first
                    ix = np.nonzero(later)[0]
                    toff = np.concatenate(([0], np.cumsum((sl["c"][ix].astype(np.int64) + 1) & ~1)))
                    d_tau = torch.zeros(int(toff[-1]) + 2, dtype=d_ar.dtype, device=self.device)
                    keep_alive.append(d_tau)
                    sl["Q"][ix] = d_tau.data_ptr() + el * toff[:-1]
                    fq = sl[ix][np.argsort(-(sl["n"][ix].astype(np.int64) * sl["c"][ix]), kind="stable")]
                    formq = (torch.from_numpy(fq.view(np.uint8).reshape(-1).copy()).to(self.device), len(fq), int(fq["n"].max()),
                             int(fq["c"].max()))
                    keep_alive.append(formq)
                t_sl = torch.from_numpy(sl.view(np.uint8).reshape(-1).copy()).to(self.device)
                keep_alive.append(t_sl)
            cp_max = [int((_cdiv(tabs_h["cp"][o: o + n]["rows"].astype(np.int64), 32)
                           * _cdiv(tabs_h["cp"][o: o + n]["cols"].astype(np.int64), 32)).max()) for o, n in CP.spans]
            self.timings["descriptors"] = self.timings.get("descriptors", 0.0) + time.perf_counter() - t_mk

            def gemm(i, st_, opA=0, alpha=1.0):
                (o, n), (t_o, t_n) = G.spans[i], tile_span[i]
                nat.check(lib.tmf_gemm_batched(self.dt, opA, alpha, 0.0, tabs_d["g"].data_ptr() + 48 * o,
                                               tabs_d["t"].data_ptr() + 16 * t_o, t_n, tile_n[i], st_), "tmf_gemm_batched")

            def copy(i, st_):
                o, n = CP.spans[i]
                nat.check(lib.tmf_copy_blocks_batched(self.dt, tabs_d["cp"].data_ptr() + 40 * o, n, cp_max[i], st_),
                          "tmf_copy_blocks_batched")

            def qr(i, st_):
                o, n = QR.spans[i]
                if use_slab:
                    nat.check(lib.tmf_house_qr_regs_batched(self.dt, t_sl.data_ptr() + 48 * o, n, qr_max[i][0], qr_max[i][1], st_),
                              "tmf_house_qr_regs_batched")
                    return
                nat.check(lib.tmf_house_qr_batched(self.dt, tabs_d["qr"].data_ptr() + 40 * o, n, qr_max[i][0], qr_max[i][1],
                                                   st_), "tmf_house_qr_batched")

            def rescale(i, st_, sweep, bond):
                if not rescaling:
                    return
                o, n = RS.spans[i]
                slot = 2 + sweep * (Ls + 1) + bond
                nat.check(lib.tmf_rescale_pow2_batched(self.dt, tabs_d["rs"].data_ptr() + 24 * o, n, d_exp.data_ptr() + 8 * sweep,
                                                       d_exp.data_ptr() + 8 * slot, st_), "tmf_rescale_pow2_batched")
            return _NS(steps1=steps1, steps2=steps2, tail=tail, jd=jd, gemm=gemm, copy=copy, qr=qr, rescale=rescale, formq=formq,
                       tabs_d=tabs_d, gc=gc, gc_bond=gc_bond)

        # The two sweeps of the parallel method start as soon as the tables of their first steps are up: the tables of the later
        # steps (and of the all-bond tail) are built while the device works on the earlier ones (host: 50 us per step pair, device:
        # 220 us).  TMF_GW_CHUNKS=0: everything first, as for the other paths.
        chunked = (self.method == "parallel" and cell is None and Ls >= 96 and os.environ.get("TMF_GW_CHUNKS", "1") != "0")
        formqs = []

        def bind(T):
            return (T.steps1, T.steps2, T.tail, T.jd, T.gemm, T.copy, T.qr, T.rescale, T.tabs_d, T.gc, T.gc_bond)

        if not chunked:
            T = make(range(Ls), range(Ls - 1, -1, -1), True)
            steps1, steps2, tail, jd, gemm, copy, qr, rescale, tabs_d, gc, gc_bond = bind(T)
            formqs.append(T.formq)
        cur = torch.cuda.current_stream(self.device)
        s1 = cur.cuda_stream
        t2 = time.perf_counter()
        if self.method == "sequential":
            # ================= rightward QR sweep, then one SVD per site on the way back =================
            for jstep, (ga, qa, ra_) in enumerate(steps1):
                gemm(ga, s1)
                qr(qa, s1)
                rescale(ra_, s1, 0, jstep + 1)
            for fq in formqs:          # the isometries of the whole sweep in one launch
                if fq is not None:
                    nat.check(lib.tmf_house_form_q_batched(self.dt, fq[0].data_ptr(), fq[1], fq[2], fq[3], s1), "tmf_house_form_q_batched")
            end = next(iter(self.sect[Ls]))
            norm = abs(complex(d_ar[Rb[Ls][end]].item()))                         # (host sync: end of sweep 1)
            if rescaling:                                    # R_Ls = (stored mantissa) 2^e
                e_tot = int(d_exp[0].item())
                self.log2_norm = float(np.log2(norm) + e_tot) if norm > 0 else -np.inf
                norm_mant, norm = norm, float(np.ldexp(norm, e_tot))       # (0.0 beyond the range of a double: see log2_norm)
            else:
                norm_mant = norm
            self.timings["sweep1"] = time.perf_counter() - t2
            if not norm_mant > 0.0 or not np.isfinite(norm_mant):
                raise ValueError("the Gutzwiller projection annihilates the state")
            t3 = time.perf_counter()
            # (the last tensor was normalised by its own QR: X of the right end stays 1)
            d_sv[So[Ls][end]] = 1.0
            d_cnt[cnt_index[(Ls, end)]] = 1
            d_sw = torch.zeros(n_sec_tot, dtype=torch.int32, device=self.device)   # sweep count of every Jacobi problem
            for gw_, cw_, qw_, qw2_, (i0, nj, pmax), gb_, gx_ in steps2:
                gemm(gw_, s1)
                copy(cw_, s1)
                qr(qw_, s1)
                qr(qw2_, s1)
                nat.check(lib.tmf_jacobi_compact_batched(self.dt, tabs_d["jc"].data_ptr() + 64 * i0, nj, pmax,
                                                         d_sw.data_ptr() + 4 * i0, s1),
                          "tmf_jacobi_compact_batched")
                gemm(gb_, s1)
                gemm(gx_, s1)
            torch.cuda.synchronize(self.device)
            self.timings["sweep2"] = time.perf_counter() - t3
            h = d_sw.cpu().numpy()
            nat.check_jacobi_sweeps(h, "Jacobi SVD of the canonicalisation sweep (npc.svd in canonical_form_finite)")
            if os.environ.get("TMF_JACOBI_SWEEPS"):     # development aid
                big = jd["p"] >= 0.8 * jd["p"].max()
                print(f"[gutzwiller] Jacobi sweeps: all problems mean {h.mean():.1f} max {h.max()}; p >= {int(0.8 * jd['p'].max())}: "
                      f"mean {h[big].mean():.1f} hist {np.bincount(h[big]).tolist()}", flush=True)
        else:
            # ================= two QR-only sweeps, independent of each other, on two HIP streams =================
            # (Two streams taken from the pool one after the other: streams share the device's few hardware queues round robin,
            # and a side stream next to the CURRENT stream landed on the current stream's queue in one call out of four or
            # five - the sweeps then ran one after the other, 220 instead of 114 ms at config 5.)
            first, side = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)
            first.wait_stream(cur)
            side.wait_stream(cur)
            sa, s2 = first.cuda_stream, side.cuda_stream
            def issue(T_, first_step):
                for jstep, ((ga, qa, ra_), (gb_, cb_, qb, rb_)) in enumerate(zip(T_.steps1, T_.steps2), first_step):     # interleaved issue: both queues stay fed
                    T_.gemm(ga, sa)
                    T_.qr(qa, sa)
                    T_.rescale(ra_, sa, 0, jstep + 1)               # R_{j+1}
                    T_.gemm(gb_, s2)
                    T_.copy(cb_, s2)
                    T_.qr(qb, s2)
                    T_.rescale(rb_, s2, 1, Ls - 1 - jstep)          # L_j, j = Ls - 1 - jstep

            if chunked:
                bounds = [b_ for b_ in (0, 16, 64, 224) if b_ < Ls] + [Ls]
                for a_, b_ in zip(bounds[:-1], bounds[1:]):
                    Tc = make(range(a_, b_), range(Ls - 1 - a_, Ls - 1 - b_, -1), False)
                    issue(Tc, a_)
                    formqs.append(Tc.formq)
                T = make((), (), True)
                steps1, steps2, tail, jd, gemm, copy, qr, rescale, tabs_d, gc, gc_bond = bind(T)
            else:
                issue(T, 0)
            cur.wait_stream(first)
            cur.wait_stream(side)
            bond_exp = None
            if cell is None:
                norm = abs(complex(d_ar[Lb[0][next(iter(self.sect[0]))]].item()))   # (host sync: both sweeps done)
                if rescaling:
                    h_exp = d_exp.cpu().numpy()
                    e_tot = int(h_exp[1])                    # L_0 = (stored mantissa) 2^e_tot
                    eR = np.concatenate(([0], h_exp[2 + 1: 2 + Ls + 1]))                       # bond 0: R = 1, never scaled
                    eL = np.concatenate((h_exp[2 + (Ls + 1): 2 + (Ls + 1) + Ls], [0]))        # bond Ls: L = 1
                    bond_exp = (eR + eL - e_tot).astype(np.int64)          # C_j = R_j' L_j' 2^bond_exp[j] / mantissa
                    self.log2_norm = float(np.log2(norm) + e_tot) if norm > 0 else -np.inf
                    norm_mant, norm = norm, float(np.ldexp(norm, e_tot))   # (0.0 beyond the range of a double: see log2_norm)
            else:
                norm = float(np.sqrt(cell["eta"]))      # norm of the projected state per unit cell
            self.timings["sweeps"] = time.perf_counter() - t2
            if not bool(torch.isfinite(torch.view_as_real(d_ar) if self.cplx else d_ar).all()):
                raise FloatingPointError("non-finite entries after the QR sweeps of the projected MPS")
            if os.environ.get("TMF_GW_DEBUG"):      # development aid: the triangular factors of both sweeps, per bond and sector
                h_all = d_ar.cpu().numpy()
                self.debug = {"R": {(j, c): h_all[Rb[j][c]: Rb[j][c] + n * n].reshape(n, n).T.copy() for j in range(Ls + 1)
                                    for c, n in self.sect[j].items()},
                              "L": {(j, c): h_all[Lb[j][c]: Lb[j][c] + n * n].reshape(n, n).T.copy() for j in range(Ls + 1)
                                    for c, n in self.sect[j].items()}, "sect": [dict(s_) for s_ in self.sect]}
            if bond_exp is None:
                norm_mant = norm
            if not norm_mant > 0.0 or not np.isfinite(norm_mant):
                raise ValueError("the Gutzwiller projection annihilates the state")
            # ================= every bond at once =================
            t3 = time.perf_counter()
            if bond_exp is None or not bond_exp.any():
                gemm(tail[0], s1, alpha=1.0 / norm_mant)
            else:      # bonds whose two factors were scaled by a different total than the norm: one launch per power of two
                ks = bond_exp[np.asarray(gc_bond)]
                for kv in np.unique(ks):
                    self._gemm_now([gc[i] for i in np.nonzero(ks == kv)[0]], 0, s1, keep_alive, alpha=float(np.ldexp(1.0 / norm_mant, int(kv))))
            qr(tail[3], s1)
            for fq in formqs:
                if fq is not None:
                    nat.check(lib.tmf_house_form_q_batched(self.dt, fq[0].data_ptr(), fq[1], fq[2], fq[3], s1), "tmf_house_form_q_batched")
            d_sw = torch.zeros(n_sec_tot, dtype=torch.int32, device=self.device)
            nat.check(lib.tmf_jacobi_compact_batched(self.dt, tabs_d["jc"].data_ptr(), n_sec_tot, int(jd["p"].max()),
                                                     d_sw.data_ptr(), s1), "tmf_jacobi_compact_batched")
            if cell is not None:
                # The bond that closes the cell must carry the SAME basis on both of its copies.  On the last bond the SVD of
                # C_Ls = R_Ls / sqrt(eta) just taken stands: R_Ls is diag(lam) of _close_cell pushed through the cell once more
                # in factored form, which reproduces the Schmidt values of the fixed point and damps what the Gram matrices
                # put there by rounding (directions the projected cell annihilates came out of them at 1e-7 .. 1e-6).  Bond 0
                # takes values and count from there, and its basis is that one carried through the unitary the leftward
                # sweep arrived with: L_0 Vz_0 = sqrt(eta) L_Ls Vz_Ls with L_Ls = 1, i.e. Vz_0 = L_0^H Vz_Ls / sqrt(eta).
                g0 = []
                for q, n in self.sect[0].items():
                    ql = self.end_of[q]
                    g0.append((P(Lb[0][q]), P(Vzb[Ls][ql]), P(Vzb[0][q]), n, n, n, n, n, n))
                    d_sv[So[0][q]: So[0][q] + n] = d_sv[So[Ls][ql]: So[Ls][ql] + n]
                    d_cnt[cnt_index[(0, q)]] = d_cnt[cnt_index[(Ls, ql)]]
                self._gemm_now(g0, 1, s1, keep_alive, alpha=1.0 / norm)
            gemm(tail[1], s1)
            gemm(tail[2], s1, opA=1)
            torch.cuda.synchronize(self.device)
            nat.check_jacobi_sweeps(d_sw.cpu().numpy(), "Jacobi SVD of the bond matrices (npc.svd in canonical_form_finite)")
            # Every bond was truncated on its own, so a row of B_j whose Schmidt value s sits within a few decades of the
            # cutoff has lost (cutoff / s)^2 of its norm to dropped columns.  One batched Gram-Schmidt pass (Cholesky-QR
            # twice: the Gram matrices are close to 1) over the kept columns of every B_j^H restores the isometry exactly;
            # its triangular factor has a positive diagonal, so the basis of bond j moves by <= (cutoff / s)^2 per index
            # and the state by less than the truncation itself (no compensation in B_{j-1}).
            h_cnt0 = d_cnt.cpu().numpy()
            ob, orows, ocend = [], [], []
            for j in range(Ls):
                for c, v in Winfo[j].items():
                    k = int(h_cnt0[cnt_index[(j, c)]])
                    if k > 0:
                        ob.append(P(Bho[j][c]))
                        orows.append(v[1])
                        ocend.append(k)
            self._orthonormalise_columns(ob, orows, ocend, s1, keep_alive)
            torch.cuda.synchronize(self.device)
            self.timings["svd+gauge"] = time.perf_counter() - t3
        # ================= results =================
        t4 = time.perf_counter()
        h_sv, h_cnt = d_sv.cpu().numpy(), d_cnt.cpu().numpy()
        if not np.isfinite(h_sv).all():      # (never silently: a NaN in a factorisation would otherwise just thin out the bonds)
            raise FloatingPointError("non-finite Schmidt values in the canonicalisation of the projected MPS")
        b_lo = min(o for bo in Bho for o in bo.values())
        b_hi = max(Bho[j][c] + v[1] * self.sect[j][c] for j in range(Ls) for c, v in Winfo[j].items())
        # (into page-locked memory: the pageable copy of .cpu() moved the 0.3 GB of config 5 at 15 GB/s; the blocks of the result are
        # views of this buffer and keep it alive, torch's host allocator hands a freed one out again)
        try:
            t_b = torch.empty(b_hi - b_lo, dtype=d_ar.dtype, pin_memory=True)
        except RuntimeError:          # (no page-locked memory of that size to be had: the pageable copy)
            t_b = None
        if t_b is not None:
            t_b.copy_(d_ar[b_lo: b_hi], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            h_b = t_b.numpy()
        else:
            h_b = d_ar[b_lo: b_hi].cpu().numpy()
        cnt = [{c: int(h_cnt[cnt_index[(j, c)]]) for c in self.sect[j]} for j in range(Ls + 1)]
        lam, ch, offs = [], [], []
        for j in range(Ls + 1):
            o, oo, ll, cc = 0, {}, [], []
            for c in self.sect[j]:
                k = cnt[j][c]
                oo[c] = o
                o += k
                ll.append(h_sv[So[j][c]: So[j][c] + k])
                cc.append(np.full(k, c, np.int64))
            sv = np.concatenate(ll) if ll else np.zeros(0)
            lam.append(sv / np.linalg.norm(sv))
            ch.append(np.concatenate(cc) if cc else np.zeros(0, np.int64))
            offs.append(oo)
        blocks = []
        for j in range(Ls):
            bl = []
            for c, v in Winfo[j].items():
                n, w, k = self.sect[j][c], v[1], cnt[j][c]
                if k == 0:
                    continue
                Bh = h_b[Bho[j][c] - b_lo: Bho[j][c] - b_lo + w * n].reshape(n, w)      # [k, (sg, c', a)] = Bh^T
                for (sg, cp_), c0 in v[2].items():
                    kr = cnt[j + 1][cp_]
                    if kr == 0:
                        continue
                    a_ = Bh[:k, c0: c0 + kr]
                    bl.append((sg, c, cp_, offs[j][c], offs[j][c] + k, offs[j + 1][cp_], offs[j + 1][cp_] + kr,
                               a_.conj() if self.cplx else a_))
            blocks.append(bl)
        self.timings["download"] = time.perf_counter() - t4
        del keep_alive
        return blocks, lam, ch, norm

    # ---- infinite MPS: closing the unit cell ----------------------------------------------------------
    def _gemm_table(self, G, keep):
        """Uploads the descriptor table of the launches collected in `G`; returns run(i, stream, opA, alpha)."""
        torch, lib = self.torch, self.lib
        gt = G.table()
        tiles, span, tile_n, to = [], [], [], 0
        for (o, n) in G.spans:
            tl, tn = _gemm_tiles(gt[o: o + n])
            tiles.append(tl)
            span.append((to, len(tl)))
            tile_n.append(tn)
            to += len(tl)
        tiles = np.concatenate(tiles) if tiles else np.zeros((0, 4), np.int32)
        d_g = torch.from_numpy(gt.view(np.uint8).reshape(-1).copy() if gt.size else np.zeros(48, np.uint8)).to(self.device)
        d_t = torch.from_numpy(tiles.reshape(-1).copy() if tiles.size else np.zeros(4, np.int32)).to(self.device)
        keep += [d_g, d_t]

        def run(i, st, opA=0, alpha=1.0):
            (o, n), (t_o, t_n) = G.spans[i], span[i]
            if t_n:
                nat.check(lib.tmf_gemm_batched(self.dt, opA, alpha, 0.0, d_g.data_ptr() + 48 * o, d_t.data_ptr() + 16 * t_o,
                                               t_n, tile_n[i], st), "tmf_gemm_batched")
        return run

    def _close_cell(self, c):
        """Boundary of the unit cell of an infinite MPS (what TeNPy's ``MPS.canonical_form_infinite1`` obtains from the
        dominant eigenvectors of its ``TransferMatrix``, called at gutzwiller.py:272 / :475).

        The projected cell tensors T_0 .. T_{Ls-1} define the transfer maps l -> sum_s T_s^H l T_s (left Gram matrix,
        pushed through the cell to the right) and r -> sum_s T_s r T_s^H (right Gram matrix, pushed to the left).  Their
        dominant eigenvectors on the bond that closes the cell are found by Arnoldi iteration: the Krylov bookkeeping runs
        on the host, every application of the cell is 2 Ls (resp. 3 Ls) batched launches over all charge sectors.  With
        l = U_l D_l U_l^H, r = U_r D_r U_r^H (Jacobi, one launch), M = D_l^1/2 U_l^H U_r D_r^1/2 = U S V^H (Jacobi), the gauge
        G = U_r D_r^1/2 V on the last tensor and its pseudo-inverse V^H D_r^-1/2 U_r^H on the first make the right environment
        of that bond the identity and the left one diag(S^2): from there the cell is a finite chain between known
        boundaries, which the two QR sweeps and the per-bond SVDs of the finite algorithm canonicalise."""
        torch, lib, Ls, el = self.torch, self.lib, self.Ls, self.elem
        P, d_ar, Toff, Rb, Lb, stream, keep = c.P, c.d_ar, c.Toff, c.Rb, c.Lb, c.stream, c.keep
        sec0 = list(self.sect[0])
        secL = [self.end_of[q] for q in sec0]
        n0 = np.array([self.sect[0][q] for q in sec0], np.int64)
        if int(n0.max()) > 4096:      # (tmf_jacobi_compact_batched; beyond ~100 states per sector its columns live in global memory)
            raise NotImplementedError(f"a charge sector of the cell boundary holds {int(n0.max())} > 4096 states")
        nsec = len(sec0)
        sq_off = np.concatenate(([0], np.cumsum((n0 * n0 + 1) & ~1)))
        tdt = d_ar.dtype

        def room(count):
            t = torch.zeros(int(count) + 2, dtype=tdt, device=self.device)
            keep.append(t)
            return t

        # ---- merged copies of the tensors: (s, c) rows x c' (left-merged) and ((s, c') x c)^H (right-merged) -----------
        TVm, TWh, cp = {}, {}, []
        for j in range(Ls):
            for cp_, (voff, m, rows) in c.Vinfo[j].items():
                TVm[(j, cp_)] = room(m * self.sect[j + 1][cp_]).data_ptr()
            for q, v in c.Winfo[j].items():
                TWh[(j, q)] = room(v[1] * self.sect[j][q]).data_ptr()
            for x in self.sb[j]:
                _, sg, q, cp_, *_ = x
                n, npr = self.sect[j][q], self.sect[j + 1][cp_]
                m, rows = c.Vinfo[j][cp_][1], c.Vinfo[j][cp_][2]
                w, cols = c.Winfo[j][q][1], c.Winfo[j][q][2]
                src = P(Toff[(j, sg, q)])
                cp.append((src, TVm[(j, cp_)] + el * rows[(sg, q)], n, npr, n, m, 0, 0))
                cp.append((src, TWh[(j, q)] + el * cols[(sg, cp_)], n, npr, n, w, 3 if self.cplx else 1, 0))
        self._copy(np.array(cp, nat.copy_desc), stream, keep)
        # ---- launch tables of the two transfer maps ---------------------------------------------------------------
        G, CPY = _Launches(nat.gemm_desc), _Launches(nat.copy_desc)
        left, right = [], []
        for j in range(Ls):
            ga, gb = [], []
            for x in self.sb[j]:
                _, sg, q, cp_, *_ = x
                voff, m, rows = c.Vinfo[j][cp_]
                n, npr = self.sect[j][q], self.sect[j + 1][cp_]
                ga.append((P(Rb[j][q]), P(Toff[(j, sg, q)]), P(voff + rows[(sg, q)]), n, npr, n, n, n, m))
            for cp_, (voff, m, rows) in c.Vinfo[j].items():
                npr = self.sect[j + 1][cp_]
                gb.append((TVm[(j, cp_)], P(voff), P(Rb[j + 1][cp_]), npr, npr, m, m, m, npr))
            left.append((G.add(_gemm_recs(ga)), G.add(_gemm_recs(gb))))
        for j in range(Ls - 1, -1, -1):
            ga, gb, cpy, wl, o = [], [], [], {}, 0
            for q, v in c.Winfo[j].items():
                wl[q] = o
                o += (self.sect[j][q] * v[1] + 1) & ~1
            for x in self.sb[j]:
                _, sg, q, cp_, *_ = x
                n, npr = self.sect[j][q], self.sect[j + 1][cp_]
                cols = c.Winfo[j][q][2]
                ga.append((P(Toff[(j, sg, q)]), P(Lb[j + 1][cp_]), P(c.Wo + wl[q] + n * cols[(sg, cp_)]), n, npr, npr, n, npr, n))
            for q, v in c.Winfo[j].items():
                n, w = self.sect[j][q], v[1]
                cpy.append((P(c.Wo + wl[q]), P(c.Yq[j][q]), n, w, n, w, 3 if self.cplx else 1, 0))
                gb.append((P(c.Yq[j][q]), TWh[(j, q)], P(Lb[j][q]), n, n, w, w, w, n))
            right.append((G.add(_gemm_recs(ga)), CPY.add(np.array(cpy, nat.copy_desc)), G.add(_gemm_recs(gb))))
        gemm = self._gemm_table(G, keep)
        stage = room(sq_off[-1])
        sc = {}
        for name, dst_tab, dst_sec, src_tab, src_sec in (("l", Rb[0], sec0, Rb[Ls], secL), ("r", Lb[Ls], secL, Lb[0], sec0)):
            put, get = np.zeros(nsec, nat.copy_desc), np.zeros(nsec, nat.copy_desc)
            for i in range(nsec):
                n = int(n0[i])
                put[i] = (stage.data_ptr() + el * int(sq_off[i]), P(dst_tab[dst_sec[i]]), n, n, n, n, 0, 0)
                get[i] = (P(src_tab[src_sec[i]]), stage.data_ptr() + el * int(sq_off[i]), n, n, n, n, 0, 0)
            sc[name] = (CPY.add(put), CPY.add(get))
        d_cp = torch.from_numpy(CPY.table().view(np.uint8).reshape(-1).copy()).to(self.device)
        keep.append(d_cp)
        cpt = CPY.table()
        cp_max = [int((_cdiv(cpt[o: o + n]["rows"].astype(np.int64), 32) * _cdiv(cpt[o: o + n]["cols"].astype(np.int64), 32)).max())
                  for o, n in CPY.spans]

        def copy(i):
            o, n = CPY.spans[i]
            nat.check(lib.tmf_copy_blocks_batched(self.dt, d_cp.data_ptr() + 40 * o, n, cp_max[i], stream), "tmf_copy_blocks_batched")

        self.cell_applications = 0

        def apply(which, vec):
            self.cell_applications += 1
            stage[: vec.size].copy_(torch.from_numpy(np.ascontiguousarray(vec, self.np_dt)))
            copy(sc[which][0])
            if which == "l":
                for a, b in left:
                    gemm(a, stream)
                    gemm(b, stream, opA=1)
            else:
                for a, cc, b in right:
                    gemm(a, stream)
                    copy(cc)
                    gemm(b, stream, opA=1)
            copy(sc[which][1])
            return stage[: vec.size].cpu().numpy()

        v0 = np.zeros(int(sq_off[-1]), self.np_dt)
        for i in range(nsec):
            n = int(n0[i])
            v0[sq_off[i]: sq_off[i] + n * n] = np.eye(n).reshape(-1)
        eta_l, lv = _dominant_eigenpair(lambda v: apply("l", v), v0)
        eta_r, rv = _dominant_eigenpair(lambda v: apply("r", v), v0)
        eta = 0.5 * (eta_l + eta_r)
        if not (abs(eta_l - eta_r) <= 1e-8 * abs(eta) and abs(eta.imag) <= 1e-8 * abs(eta) and eta.real > 0):
            raise ValueError(f"the transfer matrix of the projected unit cell has no unique positive dominant eigenvalue "
                             f"(left {eta_l}, right {eta_r}): the projected state vanishes or is not injective")
        eta = float(eta.real)

        def gram(vec, what):
            out, tr = [], 0.0
            for i in range(nsec):
                n = int(n0[i])
                out.append(vec[sq_off[i]: sq_off[i] + n * n].reshape(n, n, order="F"))
                tr = tr + np.trace(out[-1])
            out = [b / tr for b in out]                      # fixes the phase as well: a Gram matrix has a positive trace
            herm = max(np.abs(b - b.conj().T).max() for b in out)
            if herm > 1e-7 * max(np.abs(b).max() for b in out):
                raise ValueError(f"the dominant {what} eigenvector of the transfer matrix is not Hermitian ({herm:.1e}): "
                                 "the projected infinite MPS is not injective")
            return [0.5 * (b + b.conj().T) for b in out]

        lb, rb = gram(lv, "left"), gram(rv, "right")
        tau = float(sum(np.sum(a * b.T).real for a, b in zip(lb, rb)))        # tr(l r) = sum of the squared Schmidt weights
        # ---- eigen-decompositions of l and r: one Jacobi launch -----------------------------------------------------
        X = room(2 * sq_off[-1])
        Uv = room(2 * sq_off[-1])
        for k_, bl in enumerate((lb, rb)):
            h = np.zeros(int(sq_off[-1]), self.np_dt)
            for i, b in enumerate(bl):
                h[sq_off[i]: sq_off[i] + b.size] = b.reshape(-1, order="F")
            X[k_ * int(sq_off[-1]): (k_ + 1) * int(sq_off[-1])].copy_(torch.from_numpy(h))
        s_off = np.concatenate(([0], np.cumsum(n0)))
        d_s = torch.zeros(2 * int(s_off[-1]) + 1, dtype=torch.float64, device=self.device)
        d_c = torch.zeros(2 * nsec, dtype=torch.int32, device=self.device)
        d_sw = torch.zeros(2 * nsec, dtype=torch.int32, device=self.device)
        jd = np.zeros(2 * nsec, nat.jacobi_desc)
        for k_ in range(2):
            for i in range(nsec):
                n, o = int(n0[i]), el * (k_ * int(sq_off[-1]) + int(sq_off[i]))
                jd[k_ * nsec + i] = (X.data_ptr() + o, 0, Uv.data_ptr() + o, d_s.data_ptr() + 8 * (k_ * int(s_off[-1]) + int(s_off[i])),
                                     d_c.data_ptr() + 4 * (k_ * nsec + i), _INF_GRAM_EPS ** 2, n, n, n, n)
        d_jd = torch.from_numpy(jd.view(np.uint8).reshape(-1).copy()).to(self.device)
        nat.check(lib.tmf_jacobi_compact_batched(self.dt, d_jd.data_ptr(), 2 * nsec, int(n0.max()), d_sw.data_ptr(), stream),
                  "tmf_jacobi_compact_batched")
        nat.check_jacobi_sweeps(d_sw.cpu().numpy(), "Jacobi eigen-decomposition of the transfer-matrix fixed points")
        h_s, h_c = d_s.cpu().numpy(), d_c.cpu().numpy()
        # (a Gram matrix is positive: its singular values are its eigenvalues and sum to its trace)
        for k_, what in enumerate(("left", "right")):
            tot = float(h_s[k_ * int(s_off[-1]): (k_ + 1) * int(s_off[-1])].sum())
            if abs(tot - 1.0) > 1e-6:
                raise ValueError(f"the dominant {what} eigenvector of the transfer matrix is not positive (sum of |eigenvalues| = "
                                 f"{tot:.9f} for trace 1): the projected infinite MPS is not injective")
        # ---- M = D_l^1/2 U_l^H U_r D_r^1/2 and its SVD ---------------------------------------------------------------
        Dh = np.zeros(3 * int(sq_off[-1]), self.np_dt)
        for i in range(nsec):
            n = int(n0[i])
            for k_, (src, pw) in enumerate(((0, 0.5), (1, 0.5), (1, -0.5))):
                ev = h_s[src * int(s_off[-1]) + int(s_off[i]): src * int(s_off[-1]) + int(s_off[i]) + n]
                kk = int(h_c[src * nsec + i])
                d = np.zeros(n)
                d[:kk] = ev[:kk] ** pw
                Dh[k_ * int(sq_off[-1]) + int(sq_off[i]): k_ * int(sq_off[-1]) + int(sq_off[i]) + n * n] = np.diag(d).reshape(-1)
        D = room(Dh.size)
        D[: Dh.size].copy_(torch.from_numpy(Dh))
        A1, Y, Yi, M, Wk, Vs, Gm, Hm = (room(sq_off[-1]) for _ in range(8))
        at = lambda t, i, k_=0: t.data_ptr() + el * (k_ * int(sq_off[-1]) + int(sq_off[i]))      # noqa: E731
        it = []
        for i in range(nsec):
            n = int(n0[i])
            it += [(at(Uv, i, 0), at(D, i, 0), at(A1, i), n, n, n, n, n, n), (at(Uv, i, 1), at(D, i, 1), at(Y, i), n, n, n, n, n, n),
                   (at(Uv, i, 1), at(D, i, 2), at(Yi, i), n, n, n, n, n, n)]
        self._gemm_now(it, 0, stream, keep)
        self._gemm_now([(at(A1, i), at(Y, i), at(M, i), int(n0[i]), int(n0[i]), int(n0[i]), int(n0[i]), int(n0[i]), int(n0[i]))
                        for i in range(nsec)], 1, stream, keep)
        jd = np.zeros(nsec, nat.jacobi_desc)
        for i in range(nsec):
            n = int(n0[i])
            jd[i] = (at(M, i), at(Wk, i), at(Vs, i), d_s.data_ptr() + 8 * int(s_off[i]), d_c.data_ptr() + 4 * i,
                     _INF_SCHMIDT_EPS ** 2 * tau, n, n, n, n)
        d_jd2 = torch.from_numpy(jd.view(np.uint8).reshape(-1).copy()).to(self.device)
        nat.check(lib.tmf_jacobi_compact_batched(self.dt, d_jd2.data_ptr(), nsec, int(n0.max()), d_sw.data_ptr(), stream),
                  "tmf_jacobi_compact_batched")
        nat.check_jacobi_sweeps(d_sw[:nsec].cpu().numpy(), "Jacobi SVD of the boundary bond of the unit cell")
        h_s, h_c = d_s.cpu().numpy(), d_c.cpu().numpy()
        keep += [d_jd, d_jd2, d_s, d_c, d_sw]
        count = {q: int(h_c[i]) for i, q in enumerate(sec0)}
        lam = {q: h_s[int(s_off[i]): int(s_off[i]) + count[q]].copy() for i, q in enumerate(sec0)}
        tot = np.sqrt(sum(float((v ** 2).sum()) for v in lam.values()))
        if not tot > 0:
            raise ValueError("the Gutzwiller projection annihilates the state")
        lam = {q: v / tot for q, v in lam.items()}
        # ---- gauge matrices, boundary tensors, boundary values of the two sweeps --------------------------------------
        it = []
        for i in range(nsec):
            n = int(n0[i])
            it += [(at(Y, i), at(Vs, i), at(Gm, i), n, n, n, n, n, n), (at(Yi, i), at(Vs, i), at(Hm, i), n, n, n, n, n, n)]
        self._gemm_now(it, 0, stream, keep)
        idx0 = {q: i for i, q in enumerate(sec0)}
        it = []
        for x in self.sb[Ls - 1]:
            _, sg, q, cp_, *_ = x
            n, npr = self.sect[Ls - 1][q], self.sect[Ls][cp_]
            it.append((P(Toff[(Ls - 1, sg, q)]), at(Gm, idx0[self.shift(cp_)]), P(c.Tping[(Ls - 1, sg, q)]), n, npr, npr, n, npr, n))
        self._gemm_now(it, 0, stream, keep)
        for x in self.sb[Ls - 1]:
            Toff[(Ls - 1, x[1], x[2])] = c.Tping[(Ls - 1, x[1], x[2])]
        it = []
        for x in self.sb[0]:
            _, sg, q, cp_, *_ = x
            n, npr = self.sect[0][q], self.sect[1][cp_]
            it.append((at(Hm, idx0[q]), P(Toff[(0, sg, q)]), P(c.Tpong[(0, sg, q)]), n, npr, n, n, n, n))
        self._gemm_now(it, 1, stream, keep)
        for x in self.sb[0]:
            Toff[(0, x[1], x[2])] = c.Tpong[(0, x[1], x[2])]
        for i, q in enumerate(sec0):
            n, k = int(n0[i]), count[q]
            e = np.zeros((n, n), self.np_dt)
            e[np.arange(k), np.arange(k)] = lam[q]
            d_ar[Rb[0][q]: Rb[0][q] + n * n] = torch.from_numpy(e.reshape(-1)).to(self.device)
            e[np.arange(k), np.arange(k)] = 1.0
            d_ar[Lb[Ls][secL[i]]: Lb[Ls][secL[i]] + n * n] = torch.from_numpy(e.reshape(-1)).to(self.device)
        return {"eta": eta, "lam": lam, "count": count}

    # ---- helpers ----------------------------------------------------------------------------------
    def _copy(self, recs, stream, keep):
        if len(recs) == 0:
            return
        t = self.torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(self.device)
        keep.append(t)
        mt = int((_cdiv(recs["rows"].astype(np.int64), 32) * _cdiv(recs["cols"].astype(np.int64), 32)).max())
        nat.check(self.lib.tmf_copy_blocks_batched(self.dt, t.data_ptr(), len(recs), mt, stream), "tmf_copy_blocks_batched")

    def _orthonormalise_columns(self, base, rows, c_end, stream, keep):
        """Columns [0, c_end) of every matrix base[i] (rows[i] x c_end[i], leading dimension rows[i]) orthonormalised in
        column order by tmf_bcgs_batched (two passes, Cholesky-QR panels: positive diagonal of the triangular factor)."""
        base, rows, c_end = (np.asarray(x, np.int64) for x in (base, rows, c_end))
        if base.size == 0:
            return
        order = np.argsort(-rows, kind="stable")
        base, rows, c_end = base[order], rows[order], c_end[order]
        torch = self.torch
        noff = np.concatenate(([0], np.cumsum(c_end)))[:-1]
        d_nrm = torch.zeros(int(c_end.sum()) + 1, dtype=torch.float64, device=self.device)
        soff = np.concatenate(([0], np.cumsum(c_end * 16)))[:-1]
        d_scr = torch.zeros(int((c_end * 16).sum()) + 2, dtype=torch.complex128 if self.cplx else torch.float64, device=self.device)
        nd = np.zeros(base.size, nat.norms_desc)
        nd["src"], nd["out"] = base, d_nrm.data_ptr() + 8 * noff
        nd["n"], nd["c"], nd["lds_"] = rows, c_end, rows
        t_nd = torch.from_numpy(nd.view(np.uint8).reshape(-1).copy()).to(self.device)
        nat.check(self.lib.tmf_column_norms_batched(self.dt, t_nd.data_ptr(), base.size, stream), "tmf_column_norms_batched")
        bd = np.zeros(base.size, nat.bcgs_desc)
        bd["base"], bd["scratch"], bd["norms"] = base, d_scr.data_ptr() + self.elem * soff, nd["out"]
        bd["rows"], bd["ld"], bd["c_begin"], bd["c_end"] = rows, rows, 0, c_end
        t_bd = torch.from_numpy(bd.view(np.uint8).reshape(-1).copy()).to(self.device)
        wb = int(self.lib.tmf_bcgs_work_bytes(nat._p(bd), base.size))
        d_work = torch.empty(max(wb, 16), dtype=torch.uint8, device=self.device)
        keep += [d_nrm, d_scr, t_nd, t_bd, d_work]
        nat.check(self.lib.tmf_bcgs_batched(self.dt, t_bd.data_ptr(), nat._p(bd), base.size, 2, 1, d_work.data_ptr(), wb, stream),
                  "tmf_bcgs_batched")

    def _gemm_now(self, items, opA, stream, keep, alpha=1.0):
        if not items:
            return
        d = _gemm_recs(items)
        tiles, tn = _gemm_tiles(d)
        td = self.torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(self.device)
        tt = self.torch.from_numpy(tiles.reshape(-1).copy()).to(self.device)
        keep += [td, tt]
        nat.check(self.lib.tmf_gemm_batched(self.dt, opA, alpha, 0.0, td.data_ptr(), tt.data_ptr(), len(tiles), tn, stream),
                  "tmf_gemm_batched")


# ---------------------------------------------------------------------------------------------------
# public entry points
# ---------------------------------------------------------------------------------------------------
def _check_unit_cell_width(mps, unit_cell_width, group=2):
    """gutzwiller.py:73-88."""
    mps = native(mps)
    if unit_cell_width is None:
        unit_cell_width = mps.unit_cell_width
        if (mps.L // group) % unit_cell_width != 0:
            warn(f"Input MPS {unit_cell_width = } does not divide new MPS size {mps.L // group}\n"
                 "Default to chain geometry")
            unit_cell_width = mps.L // group
    elif (mps.L // group) % unit_cell_width != 0:
        raise ValueError(f"{unit_cell_width = } does not divide new MPS size {mps.L // group}")
    return unit_cell_width


def _drop_dead_boundary_states(blocks, lam, ch):
    """Infinite cell: a state of a bond whose row of the tensor to its right vanished (the batched Gram-Schmidt found it
    dependent on stronger ones: a direction the Gram matrices of the fixed points put there by rounding, weight < 1e-12)
    is removed from that bond - on the closing bond from both of its copies.  The rows of the tensor to the left lose the
    removed column (<= 1e-7 of their norm, measured) and are rescaled to unit length."""
    blocks, lam, ch = list(blocks), list(lam), list(ch)
    Ls = len(blocks)
    for j in range(Ls):
        n0 = len(lam[j])
        w = np.zeros(n0)
        for (_p, _ql, _qr, l0, l1, _r0, _r1, a) in blocks[j]:
            w[l0:l1] += (np.abs(a) ** 2).sum(axis=1)
        keep = w > 0.25
        if keep.all() or not keep.any():
            continue
        pos = np.cumsum(keep) - 1

        def cut(bl, rows):
            out = []
            for (p, ql, qr, l0, l1, r0, r1, a) in bl:
                lo, hi = (l0, l1) if rows else (r0, r1)
                sel = keep[lo:hi]
                if not sel.any():
                    continue
                a2 = a[sel] if rows else a[:, sel]
                n_lo = int(pos[lo + int(np.argmax(sel))])
                rng_ = (n_lo, n_lo + int(sel.sum()))
                out.append((p, ql, qr, rng_[0], rng_[1], r0, r1, a2) if rows else (p, ql, qr, l0, l1, rng_[0], rng_[1], a2))
            return out
        left = (j - 1) % Ls
        blocks[j] = cut(blocks[j], True)
        blocks[left] = cut(blocks[left], False)
        nl = len(lam[left])                      # rescale the rows of the tensor that lost a column
        w2 = np.zeros(nl)
        for (_p, _ql, _qr, l0, l1, _r0, _r1, a) in blocks[left]:
            w2[l0:l1] += (np.abs(a) ** 2).sum(axis=1)
        sc = np.where(w2 > 0.25, 1.0 / np.sqrt(np.maximum(w2, 1e-300)), 1.0)
        blocks[left] = [(p, ql, qr, l0, l1, r0, r1, a * sc[l0:l1, None]) for (p, ql, qr, l0, l1, r0, r1, a) in blocks[left]]
        new = np.asarray(lam[j])[keep]
        lam[j] = new / np.linalg.norm(new)
        ch[j] = np.asarray(ch[j])[keep]
        if j == 0:
            lam[Ls], ch[Ls] = lam[0], ch[0]
    return blocks, lam, ch


def _finish(mps, inplace, res):
    if getattr(mps, "_temfpy_amd", None) is not None:      # a TeNPy MPS came in: a TeNPy MPS goes out (gutzwiller.py:277-281)
        try:
            res = res.to_tenpy()
        except (ImportError, NotImplementedError):
            pass
    if inplace:
        mps.__class__ = res.__class__
        mps.__dict__.clear()
        mps.__dict__.update(res.__dict__)
        return None
    return res


def _total_charge(fer):
    q = np.asarray(fer.charges[fer.L])
    return int(q[0]) if q.size else 0


def abrikosov(mps, *, inplace: bool = False, return_canonical: bool = True, cutoff: float = 1e-12,
              q_left: None | int = None, unit_cell_width: int | None = None, device: str = "cuda:0",
              method: Literal["sequential", "parallel"] = "parallel"):
    """Projection from Abrikosov fermions to a spin-1/2 Hilbert space (gutzwiller.py:95-281): sites 2i, 2i+1
    hold f_up, f_down; single occupation of f_up -> up, of f_down -> down, empty and double occupation dropped.
    No charges survive.  ``mps``: a finite MPS (``MPSData`` / ``PfMPSData``) or the unit cell of an infinite one
    (``iMPS.iMPSData``; then ``q_left`` - the charge sector of the leftmost virtual leg that is kept, in the cell's own labels -
    is required, :197-206, and a :class:`SpiniMPSData` comes back)."""
    assert mps.L % 2 == 0, "Odd-length MPS cannot represent an Abrikosov fermion Hilbert space"   # :158-160
    fer = _as_fermions(mps)
    conserve = fer.conserve
    infinite = getattr(fer, "infinite", False)
    q, target = (fer.cell_charge if infinite else _total_charge(fer)), fer.L // 2
    err = f"Total charge must match number of spin sites. Got {q}, expected {target}"           # :178-187
    if conserve == "N":
        assert q == target, err
    else:
        assert q % 2 == target % 2, err + " (mod 2)"
    shift = None
    if infinite:                                                                                    # :197-206
        if q_left is None:
            raise ValueError("Must specify `q_left` for infinite MPS.")
        sectors = np.unique(fer.charges[0])
        if q_left not in sectors:
            raise ValueError(f"`q_left` must be a charge sector of the leftmost virtual leg, got {q_left = }, "
                             f"valid sectors are {sectors.tolist()}")
        shift = (lambda c: c - q) if conserve == "N" else (lambda c: (c - q) % 2)
    else:
        if q_left not in (None, 0):
            warn(f"`q_left` must be 0 for finite MPS, got {q_left = }, setting it to 0.")           # :192-196
        q_left = 0
    ucw = _check_unit_cell_width(mps, unit_cell_width)
    if conserve == "N":
        keep = lambda j, c: c == q_left + j                      # noqa: E731   number_mask(leg, q_left + idx), :236-238
    else:
        keep = lambda j, c: c % 2 == (q_left + j) % 2            # noqa: E731   parity_mask(leg, q_left + idx)
    pr = _Projector(fer, ((0, 1), (1, 0)), keep, cutoff, device, method, shift=shift)   # kept physical states in leg order 01, 10
    blocks, lam, ch, norm = pr.run(return_canonical)
    logger.info("Completed projection to spin-1/2 space. No conserved charges left.")              # :260
    if not return_canonical:
        warn("The MPS is not in canonical form after Gutzwiller projection.\nConsider setting 'return_canonical=True'")
    if infinite:
        blocks[-1] = [(p, ql, shift(qr)) + tuple(rest) for (p, ql, qr, *rest) in blocks[-1]]
        ch[-1] = ch[0]
        if return_canonical:
            blocks, lam, ch = _drop_dead_boundary_states(blocks, lam, ch)
        res = SpiniMPSData(blocks, lam, ch, None, norm, ucw, 0, timings=pr.timings)
        if not return_canonical:
            res.form = [None] * res.L
    else:
        res = SpinMPSData(blocks, lam, ch, None, norm, ucw, canonical=return_canonical, timings=pr.timings)
        res.log2_norm = getattr(pr, "log2_norm", None)      # log2 of the norm: finite where `norm` has left the range of a double
    return _finish(mps, inplace, res)


def abrikosov_ph(mps, *, inplace: bool = False, return_canonical: bool = True, cutoff: float = 1e-12, offset: int = 0,
                 parity: Literal[0, 1] = 0, unit_cell_width: int | None = None, device: str = "cuda:0",
                 method: Literal["sequential", "parallel"] = "parallel"):
    """Projection from particle-hole rotated Abrikosov fermions (gutzwiller.py:284-486): sites 2i, 2i+1 hold
    f_up, f_down^dagger; zero occupation -> down, double occupation -> up, single occupation dropped.
    Number-conserving input keeps S^z (2 S^z = number - offset - bond index, :333, :438-441).  ``mps``: a finite MPS or the
    unit cell of an infinite one (``iMPS.iMPSData``: ``parity`` selects the parity sector of the virtual legs and ``offset`` the
    label shift, :322-334; a :class:`SpiniMPSData` comes back, its last tensor carrying the cell's 2 S^z, :446-447)."""
    assert mps.L % 2 == 0, "Odd-length MPS cannot represent an Abrikosov fermion Hilbert space"   # :354-356
    fer = _as_fermions(mps)
    conserve = fer.conserve
    infinite = getattr(fer, "infinite", False)
    q = fer.cell_charge if infinite else _total_charge(fer)
    assert q % 2 == 0, f"Total fermion parity of MPS must be even, got {q}"                       # :374-376
    shift = None
    if infinite:
        shift = (lambda c: c - q) if conserve == "N" else (lambda c: (c - q) % 2)
    else:
        if parity != 0:
            warn(f"Must use even parity sector in finite MPS, ignoring {parity = }")              # :380-381
        if offset != 0 and conserve == "N":
            warn(f"Cannot offset charge of finite MPS, ignoring {offset = }")                     # :382-383
        offset = parity = 0                                                                       # :384
    ucw = _check_unit_cell_width(mps, unit_cell_width)
    keep = lambda j, c: c % 2 == parity % 2          # noqa: E731   parity_mask(leg, parity), :418-419
    pr = _Projector(fer, ((0, 0), (1, 1)), keep, cutoff, device, method, shift=shift)    # kept physical states 00, 11 = [down, up]
    blocks, lam, ch, norm = pr.run(return_canonical)
    spin = "Sz" if conserve == "N" else None
    cell = 0
    if infinite:          # the last bond is bond 0 of the next cell: its labels without the charge of one cell
        blocks[-1] = [(p, ql, shift(qr) + (fer.L // 2 if spin == "Sz" else 0)) + tuple(rest) for (p, ql, qr, *rest) in blocks[-1]]
    if spin == "Sz":     # leg charges -= offset + idx (finite: offset = 0), :438-441
        ch = [c - offset - j for j, c in enumerate(ch)]
        blocks = [[(p, ql - offset - j, qr - offset - j - 1) + tuple(rest) for (p, ql, qr, *rest) in bl] for j, bl in enumerate(blocks)]
        cell = q - fer.L // 2          # 2 S^z of one cell (the reference gauges the last tensor by qtotal - L, :446-447)
    logger.info("Completed projection to spin-1/2 space. Conserved charge is now %s", spin)       # :462-465
    if not return_canonical:
        warn("The MPS is not in canonical form after Gutzwiller projection.\nConsider setting 'return_canonical=True'")
    if infinite:
        ch[-1] = ch[0]
        if return_canonical:
            blocks, lam, ch = _drop_dead_boundary_states(blocks, lam, ch)
        res = SpiniMPSData(blocks, lam, ch, spin, norm, ucw, cell, timings=pr.timings)
        if not return_canonical:
            res.form = [None] * res.L
    else:
        res = SpinMPSData(blocks, lam, ch, spin, norm, ucw, canonical=return_canonical, timings=pr.timings)
        res.log2_norm = getattr(pr, "log2_norm", None)      # log2 of the norm: finite where `norm` has left the range of a double
    return _finish(mps, inplace, res)


__all__ = ["abrikosov", "abrikosov_ph", "SpinMPSData", "SpiniMPSData"]
