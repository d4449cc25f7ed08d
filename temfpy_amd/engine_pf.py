"""Host driver of the MI355X Pfaffian (BCS / Nambu mean-field) -> MPS sweep.

Mirrors pfaffian.C_to_MPS (pfaffian.py:1785-1921) with the same batching idea as engine.py: all
cuts and all sites of the chain are independent given the Nambu correlation matrix, so every
stage is one batched launch.  C is a projector here as well, and in the Majorana basis
1 - A = conj(A) for every diagonal block A, which gives the whole Bogoliubov matrix of a cut side
from the Slater building blocks:

  entangled pairs (lambda, 1 - lambda)   left singular vectors of the off-diagonal block, Rayleigh-Ritz
                                         (Engine.entangled_stage; 2 k_e directions above the threshold)
  filled basis  (lambda ~ 1)             dominant subspace of A, orthogonal to the entangled vectors
  empty basis   (lambda ~ 0)             complex conjugate of the filled basis (Nambu symmetry,
                                         pfaffian.py:886,891)
  vacuum parity                          det(v) = (-1)^(p + n(n-1)/2) with the LU kernel instead of the
                                         singular-value gap rule of pfaffian.py:396-456
  _pfaffian_matrix (pfaffian.py:1258-1410)  one LU / Schur pass on [[U^*, Q], [P, 0]] (nambu.hip)
  _tensor_block    (pfaffian.py:1429-1479)  gathered-Pfaffian kernel (pf_gather.hip)

Limits of this round: no eigenvalue-1/2 modes (kh = 0, i.e. no Majorana zero modes at a cut),
at most 31 entangled pairs per cut, sub-Pfaffians of order <= 32.
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np

from . import _native as nat
from .engine import Engine, P_RANGE, PANEL_W, _cdiv, _sector_list

logger = logging.getLogger("temfpy_amd.pfaffian")


def parity_n_argsort(exc):
    """pfaffian.py:986-997."""
    exc = np.asarray(exc).ravel()
    return np.lexsort((np.arange(len(exc)), exc, exc % 2))


def _bunched(x):
    idx = np.nonzero(x[1:] != x[:-1])[0]
    idx = np.concatenate(([0], idx + 1, [len(x)]))
    return {int(x[idx[i]]): (int(idx[i]), int(idx[i + 1])) for i in range(len(idx) - 1)}


class PfBond:
    """Schmidt data of one cut (pfaffian.SchmidtVectors)."""

    def __init__(self, x, e, pL, pR, sets, lam_raw):
        self.x, self.e, self.pL, self.pR = x, e, pL, pR
        self.sets = sets                      # (chi, k) bool, sorted by (parity, number)
        self.lam_raw = lam_raw
        self.lam = lam_raw / np.linalg.norm(lam_raw)
        exc = sets.sum(axis=1)
        self.idx_n = _bunched(exc)
        self.idx_parity = _bunched(exc % 2)

    @property
    def chi(self):
        return len(self.lam)

    @property
    def k(self):
        return len(self.e)

    def parity(self, which="T"):
        return self.pL if which == "L" else self.pR if which == "R" else (self.pL + self.pR) % 2

    def side_sets(self, side):
        return self.sets if side == "L" else self.sets[:, ::-1]


class PfSite:
    def __init__(self, mode, norm, qtotal, leg_idx_bra, blocks, chi_bra, chi_ket):
        self.mode, self.norm, self.qtotal = mode, norm, qtotal
        self.leg_idx_bra = leg_idx_bra
        self.blocks = blocks    # (n_bra, n_ket) -> (r0, r1, c0, c1, ndarray)
        self.chi_bra, self.chi_ket = chi_bra, chi_ket

    def dense(self):
        M = np.zeros((2 * self.chi_bra, self.chi_ket), complex)
        for (r0, r1, c0, c1, blk) in self.blocks.values():
            M[self.leg_idx_bra[r0:r1], c0:c1] = blk
        t = M.reshape(2, self.chi_bra, self.chi_ket)  # LegPipe([p, bra], sort=False): p more major
        return t if self.mode == "left" else t.transpose(0, 2, 1)


class PfMPSData:
    def __init__(self, bonds, sites, ortho_center, unit_cell_width, timings):
        self.bonds, self.sites = bonds, sites
        self.L = len(sites)
        self.ortho_center, self.unit_cell_width = ortho_center, unit_cell_width
        self.form = ["A"] * ortho_center + ["B"] * (self.L - ortho_center)
        self.timings = timings

    @property
    def lam(self):
        return [b.lam for b in self.bonds]

    @property
    def chi(self):
        return [b.chi for b in self.bonds]

    def entanglement_entropy(self, all_bonds=False):
        out = np.zeros(self.L + 1)
        for i, b in enumerate(self.bonds):
            p = b.lam**2
            p = p[p > 0]
            out[i] = -(p * np.log(p)).sum()
        return out if all_bonds else out[1:-1]

    def dense_tensors(self):
        return [s.dense() for s in self.sites]

    def to_tenpy(self, verify=True):
        """Assemble ``tenpy.networks.mps.MPS`` like pfaffian.py:1485-1489, :1628-1658, :1750-1778, :1916-1919:
        parity legs offset by the reference parity to the left of each bond, the physical leg merged into the bra leg
        by an unsorted, unbunched ``LegPipe`` (p more major), rows addressed through ``leg_idx_bra``.

        TeNPy is not installed in the build environment, so - as for the Slater path (``MPSData.to_tenpy``) - the
        assembly checks itself at run time: every assembled tensor (16 sampled sites on long chains) is read back
        with ``to_ndarray()`` and must equal this object's own dense tensor, else RuntimeError."""
        import tenpy.linalg.np_conserved as npc
        from tenpy import networks

        site = networks.site.FermionSite(conserve="parity")
        leg_p, chinfo = site.leg, site.leg.chinfo

        def legcharge(bond, qconj):      # pfaffian.py:1485-1489 with idx_parity as {parity: slice}
            idx = {(par + bond.pL) % 2: slice(a, b) for par, (a, b) in bond.idx_parity.items()}
            return npc.LegCharge.from_qdict(chinfo, idx, qconj=qconj)

        check = set(range(self.L)) if self.L <= 64 else set(np.linspace(0, self.L - 1, 16).astype(int).tolist())
        tensors = []
        for i, s in enumerate(self.sites):
            left = s.mode == "left"
            bra, ket = (self.bonds[i], self.bonds[i + 1]) if left else (self.bonds[i + 1], self.bonds[i])
            qconj = (+1, -1) if left else (-1, +1)
            names = ("vL", "vR") if left else ("vR", "vL")
            leg_bra = legcharge(bra, qconj[0])
            pipe = npc.LegPipe([leg_p, leg_bra], qconj=leg_bra.qconj, sort=False, bunch=False)
            B = npc.zeros([pipe, legcharge(ket, qconj[1])], labels=[f"(p.{names[0]})", names[1]], qtotal=(s.qtotal,),
                          dtype=complex)
            for (r0, r1, c0, c1, blk) in s.blocks.values():
                B[s.leg_idx_bra[r0:r1], slice(c0, c1)] = blk
            B = B.split_legs()
            if verify and i in check:
                got = B.to_ndarray()
                want = s.dense() if left else s.dense().transpose(0, 2, 1)
                if got.shape != want.shape or not np.array_equal(got, want):
                    raise RuntimeError(f"TeNPy assembly self-check failed at site {i} (Pfaffian path); use as_tenpy=False")
            tensors.append(B)
        psi = networks.mps.MPS([site] * self.L, tensors, self.lam, form=self.form, unit_cell_width=self.unit_cell_width)
        psi._temfpy_amd = self
        return psi


class PfEngine(Engine):
    lu_blocked_from = int(os.environ.get("TMF_PF_LU_BLOCKED", "128"))    # blocked multi-launch LU from this many pivot columns on

    def _lu_schur(self, W, det, mb, mk, k, ldw, what):
        """det(A) and the Schur complement of the leading k x k block of every matrix (in place), by the fully pivoted
        blocked LU over several launches (tmf_lu_block_batched + tmf_lu_trsm_batched + one batched MFMA GEMM per 64 pivot
        columns; 2x faster than the one-workgroup kernel on the (L + n)^2 Nambu matrices of config 4) or, for small pivot
        blocks, by the one-workgroup kernel."""
        W, det, mb, mk, k, ldw = (np.asarray(x) for x in (W, det, mb, mk, k, ldw))
        mb, mk, k, ldw = (x.astype(np.int64) for x in (mb, mk, k, ldw))
        n = len(W)
        if n == 0:
            return
        if int(k.max()) < self.lu_blocked_from:
            sd = np.zeros(n, nat.schur_desc)
            sd["W"], sd["S"], sd["det"] = W, 0, det
            sd["mb"], sd["mk"], sd["k"], sd["ldw"], sd["lds"] = mb, mk, k, ldw, 1
            t = self._up(sd)
            nat.check(self.lib.tmf_lu_schur_batched(self.dtype, t.data_ptr(), n, int(mb.max()), self.stream), what)
            return
        order = np.argsort(-k, kind="stable")             # active matrices = a prefix at every outer step
        W, det, mb, mk, k, ldw = (x[order] for x in (W, det, mb, mk, k, ldw))
        el = self.elem
        d_piv = self.torch.zeros(int(k.sum()) + 1, dtype=self.torch.int32, device=self.device)
        tk = np.where(k > 0, 64 * mk, 0)
        d_T = self._alloc(int(tk.sum()) + 1)
        self._keep += [d_piv, d_T]
        ld = np.zeros(n, nat.lublock_desc)
        ld["W"], ld["det"] = W, det
        ld["piv"] = d_piv.data_ptr() + 4 * (np.cumsum(k) - k)
        ld["T"] = d_T.data_ptr() + el * (np.cumsum(tk) - tk)
        ld["mb"], ld["mk"], ld["k"], ld["ldw"] = mb, mk, k, ldw
        t = self._up(ld)
        Wa, Ta = ld["W"].astype(np.uint64), ld["T"].astype(np.uint64)
        for j0 in range(0, max(int(k.max()), 1), 64):
            nact = int((k > j0).sum())
            nat.check(self.lib.tmf_lu_block_batched(self.dtype, t.data_ptr(), n if j0 == 0 else nact, j0, 64,
                                                    int(mb.max() if j0 == 0 else mb[:nact].max()), self.stream), what)
            if nact == 0:
                break
            cend = np.minimum(k[:nact], j0 + 64)
            nat.check(self.lib.tmf_lu_trsm_batched(self.dtype, t.data_ptr(), nact, j0, 64, int((mk[:nact] - cend).max()), self.stream), what)
            self.gemm(0, -1.0, 1.0, Wa[:nact] + ((cend + j0 * ldw[:nact]) * el).astype(np.uint64), Ta[:nact],
                      Wa[:nact] + ((cend + cend * ldw[:nact]) * el).astype(np.uint64), mb[:nact] - cend, mk[:nact] - cend, cend - j0,
                      ldw[:nact], np.full(nact, 64), ldw[:nact])

    range_floor_tol = 3e-15  # see Engine.entangled_stage_adaptive
    pf_sweep_impl = os.environ.get("TMF_PF_SWEEP", "cpp")    # "cpp": tmf_pfaffian_sweep (csrc/sweep_pf.inc); "python": run_py (A/B)

    def run(self, C, trunc, ortho_center, unit_cell_width, threads=None):
        """One Nambu correlation matrix -> MPS conversion.  The sweep runs in C++ behind ``tmf_pfaffian_sweep``; inputs whose
        cuts carry eigenvalue-1/2 modes (TMF_E_HALF_MODES: following the reference there needs SciPy's seeded ``ortho_group``
        stream) and ``TMF_PF_SWEEP=python`` take the Python orchestration of the same kernels (:meth:`run_py`)."""
        if self.pf_sweep_impl == "cpp":
            res = self.run_cpp(C, trunc, ortho_center, unit_cell_width, threads)
            if res is not None:
                return res
        return self.run_py(C, trunc, ortho_center, unit_cell_width, threads)

    def run_cpp(self, C, trunc, ortho_center, unit_cell_width, threads=None):
        import ctypes
        lib = self.lib
        t_all = time.perf_counter()
        if self._ctx is None:
            ctx = ctypes.c_void_p()
            nat.check(lib.tmf_ctx_create(self.device.index or 0, ctypes.byref(ctx)), "tmf_ctx_create")
            self._ctx = ctx
            import weakref
            self._ctx_finalizer = weakref.finalize(self, lib.tmf_ctx_destroy, ctypes.c_void_p(ctx.value))
            self._ctx_finalizer.atexit = False
        C = np.ascontiguousarray(C, np.complex128)
        L = len(C) // 2
        sectors = _sector_list(trunc, L)
        sec = None if sectors is None else np.ascontiguousarray(sectors, np.int64)
        flags = (nat.SWEEP_CHECKS if self.checks else 0) | (0 if self.filled_cholqr else nat.SWEEP_NO_CHOLQR)
        par = nat.SweepParams(L=L, chi_max=int(trunc.chi_max or 0), svd_min=float(trunc.svd_min),
                              degeneracy_tol=float(trunc.degeneracy_tol), sectors=None if sec is None else sec.ctypes.data,
                              ortho_center=int(ortho_center), site_lo=0, site_hi=L, n_sectors=0 if sec is None else int(sec.size),
                              is_complex=1, host_threads=int(threads or self.host_threads), flags=flags)
        res = ctypes.c_void_p()
        st = lib.tmf_pfaffian_sweep(self._ctx, C.ctypes.data, ctypes.byref(par), self.range_floor_tol, ctypes.byref(res))
        if st == nat.E_HALF_MODES:
            return None
        if st == -2 and "Jacobi iteration did not converge" in lib.tmf_last_error().decode():
            raise np.linalg.LinAlgError(lib.tmf_last_error().decode())
        nat.check(st, "tmf_pfaffian_sweep")
        try:
            n_out, n_chk, info = ctypes.c_int64(), ctypes.c_int32(), nat.SweepInfo()
            nat.check(lib.tmf_pf_result_dims(res, None, None, ctypes.byref(n_out), ctypes.byref(n_chk), ctypes.byref(info)), "dims")
            fl = nat.PfFlat()
            nat.check(lib.tmf_pf_result_flat(res, ctypes.byref(fl)), "tmf_pf_result_flat")

            def arr(ptr, dtype, count):          # one copy out of the library's memory per table
                if count == 0:
                    return np.zeros(0, dtype)
                nbytes = count * np.dtype(dtype).itemsize
                return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), (nbytes,)).view(dtype).copy()

            nb, nsite, nblk = int(fl.n_bonds), int(fl.n_sites), int(fl.n_blocks)
            bond = arr(fl.bond, np.int32, 4 * nb).reshape(nb, 4)
            e_off, s_off, l_off = (arr(p, np.int64, nb + 1) for p in (fl.e_off, fl.sets_off, fl.lam_off))
            e_f, s_f, l_f = arr(fl.e, np.float64, int(e_off[-1])), arr(fl.sets, np.uint8, int(s_off[-1])), arr(fl.lam_raw, np.float64, int(l_off[-1]))
            site = arr(fl.site, np.int32, 5 * nsite).reshape(nsite, 5)
            norm = arr(fl.norm, np.float64, nsite)
            leg_off, blk_off = arr(fl.leg_off, np.int64, nsite + 1), arr(fl.blk_off, np.int64, nsite + 1)
            leg = arr(fl.leg_idx_bra, np.int32, int(leg_off[-1])).astype(np.int64)
            blk = arr(fl.blk, np.int64, 7 * nblk).reshape(nblk, 7)
            # the tensors go straight from the device into page-locked memory owned by the result (no copy through the library)
            buf, keep = self.default_sink().alloc(max(int(fl.out_elems), 1) * 16)
            nat.check(lib.tmf_pf_result_download(res, buf.ctypes.data), "tmf_pf_result_download")
            flat = buf[: int(fl.out_elems) * 16].view(np.complex128)
            flat_keep = keep
            self.check_results = {}
            if n_chk.value:
                vals, cuts, kinds = np.zeros(n_chk.value), np.zeros(n_chk.value, np.int32), np.zeros(n_chk.value, np.int32)
                nat.check(lib.tmf_pf_result_checks(res, nat._p(vals), nat._p(cuts), nat._p(kinds)), "checks")
                names = ("vL is not unitary", "vL does not diagonalise C_LL", "vR is not unitary", "vR does not diagonalise C_RR")
                for kd in range(4):      # one entry per message: the worst cut
                    m = np.nonzero(kinds == kd)[0]
                    if m.size:
                        w = m[int(np.argmax(vals[m]))] if np.all(np.isfinite(vals[m])) else m[int(np.argmax(~np.isfinite(vals[m])))]
                        self.check_results[f"{names[kd]} (cut {int(cuts[w])})"] = float(vals[w])
        finally:
            lib.tmf_pf_result_free(res)

        def make_bond(b):
            k, chi, pL, pR = (int(x) for x in bond[b])
            sets = s_f[s_off[b]: s_off[b + 1]].reshape(chi, k).astype(bool)
            return PfBond(b, e_f[e_off[b]: e_off[b + 1]], pL, pR, sets, l_f[l_off[b]: l_off[b + 1]])

        def make_site(i):
            mode, qtotal, chi_b, chi_k, _ = (int(x) for x in site[i])
            blocks = {}
            for nbq, nkq, r0, r1, c0, c1, o in blk[blk_off[i]: blk_off[i + 1]].tolist():
                blocks[(nbq, nkq)] = (r0, r1, c0, c1, flat[o: o + (r1 - r0) * (c1 - c0)].reshape(r1 - r0, c1 - c0))
            return PfSite("left" if mode == 0 else "right", float(norm[i]), qtotal, leg[leg_off[i]: leg_off[i + 1]], blocks, chi_b, chi_k)

        from .mps_data import LazyList
        bonds, sites = LazyList(L + 1, make_bond), LazyList(L, make_site)
        sites._keep = flat_keep
        self.timings = {}
        for i in range(16):
            name = lib.tmf_sweep_stage_name(i).decode()
            if name:
                self.timings[name] = info.stage_ms[i] * 1e-3
        self.range_iterations_used, self.range_width, self.range_floor = int(info.range_iterations), int(info.range_width), float(info.range_floor)
        self.timings["total"] = time.perf_counter() - t_all
        return self._finish(PfMPSData(bonds, sites, ortho_center, unit_cell_width, dict(self.timings)))

    def run_py(self, C, trunc, ortho_center, unit_cell_width, threads=None):
        torch = self.torch
        t_all = time.perf_counter()
        self.timings = {}
        self.det_events, self.gemm_events = [], []
        torch.cuda.current_stream(self.device).synchronize()
        self._pin_off = 0
        self.dtype, self.elem = nat.TMF_C128, 16
        el = 16
        C = np.ascontiguousarray(C, np.complex128)
        D = len(C)            # 2 L
        L = D // 2
        oc = ortho_center
        cutoff = trunc.svd_min**2
        thr2 = cutoff * (1.0 - cutoff)
        deg_tol = trunc.degeneracy_tol

        def offsets(sizes):
            o = np.concatenate(([0], np.cumsum(sizes)))
            return o[:-1].astype(np.int64), int(o[-1])

        t0 = time.perf_counter()
        d_Crm = torch.from_numpy(C.reshape(-1)).to(self.device)
        d_C = self._alloc(D * D)
        nat.check(self.lib.tmf_transpose(self.dtype, d_Crm.data_ptr(), d_C.data_ptr(), D, self.stream), "transpose")
        Cp = d_C.data_ptr()
        self._tick("upload", t0)

        # ---- cut sides ----------------------------------------------------------------------------
        cs_b, cs_side = [], []
        for b in range(L + 1):
            if b <= oc:
                cs_b.append(b), cs_side.append(0)
            if b >= oc:
                cs_b.append(b), cs_side.append(1)
        cs_b, cs_side = np.array(cs_b), np.array(cs_side)
        ncs = len(cs_b)
        ns = np.where(cs_side == 0, cs_b, L - cs_b)     # sites of the block
        n = 2 * ns                                       # dimension of the Nambu block
        m = D - n
        blk = Cp + np.where(cs_side == 0, 0, 2 * cs_b + 2 * cs_b * D) * el
        off = Cp + np.where(cs_side == 0, 2 * cs_b * D, 2 * cs_b) * el
        cidx = {(int(b), int(s_)): i for i, (b, s_) in enumerate(zip(cs_b, cs_side))}
        centre_L, centre_R = cidx[(oc, 0)], cidx[(oc, 1)]
        doE = (n > 0) & (m > 0)
        doE[centre_R] = False

        t0 = time.perf_counter()
        st = self.entangled_stage_adaptive(D, n, m, blk, off, doE, thr2, cs_b, 2 * cs_b, cs_side, Cp)
        P, p = st["P"], st["p"]
        UEp, oS, ld1 = st["UEp"], st["oS"], st["ld1"]
        self._tick("E_entangled", t0)

        # ---- host round trip 1: pairs (lambda, 1 - lambda) --------------------------------------------
        t0 = time.perf_counter()
        h_e, h_cnt = st["h_e"], st["h_cnt"]
        ke = np.zeros(ncs, np.int64)
        kh = np.zeros(ncs, np.int64)                   # pairs of eigenvalue-1/2 modes (included in ke)
        e_cut = [np.zeros(0)] * ncs
        lam_side = [np.zeros(0)] * ncs                 # eigenvalue of every entangled column of Vt
        for i in range(ncs):
            if not doE[i]:
                continue
            cnt = int(h_cnt[i])
            lam = h_e[oS[i]: oS[i] + cnt]
            if cnt % 2 or np.abs(lam + lam[::-1] - 1.0).max(initial=0.0) > 1e-8:
                raise ValueError("Eigenvalues break Nambu symmetry")  # pfaffian.py:799-800
            ke[i] = cnt // 2
            half = np.abs(lam - 0.5) <= deg_tol              # pfaffian.py:803-805
            if half.any():
                nh = int(half.sum())
                assert nh % 2 == 0 and half[ke[i] - nh // 2: ke[i] + nh // 2].all(), \
                    "1/2 eigenvalues asymmetrical in spectrum"
                kh[i] = nh // 2
            e_cut[i] = lam[ke[i]:][::-1].copy()     # lower half, ascending (pfaffian.py:839)
            lam_side[i] = lam.copy()
        ke[centre_R] = ke[centre_L]
        kh[centre_R] = kh[centre_L]
        e_cut[centre_R] = e_cut[centre_L]
        kc_ = int(ke[centre_L])
        # centre-right columns (built below): [partners of the left LOWER modes, reversed | their conjugates]
        lam_side[centre_R] = np.concatenate(((1.0 - lam_side[centre_L][kc_:])[::-1], lam_side[centre_L][kc_:]))
        nb_ = ns - ke                                  # filled / empty basis vectors
        self._tick("host_classify", t0)

        # ---- mode blocks Vtmp = [entangled (2 ke, Ritz order: lambda descending) | filled basis] -----
        t0 = time.perf_counter()
        ncol = 2 * ke + nb_
        oV, tV = offsets(n * ncol)
        d_Vt = self._alloc(tV)
        Vt = d_Vt.data_ptr() + oV * el
        cp = doE.copy()
        from .engine import _group_order
        i_ = centre_L
        nh_ = int(ke[i_] - kh[i_])
        perm = _group_order(e_cut[i_][:nh_], deg_tol) if doE[i_] and nh_ > 1 else None
        if perm is not None:
            # left modes of the centre cut inside a group of eigenvalues closer than degeneracy_tol take the order of the
            # reference's group SVD (block_svd on the lower modes, pfaffian.py:855; the conjugates follow, :884): lower mode j
            # (ascending e) is column 2 ke - 1 - j, its conjugate column j
            k2 = 2 * int(ke[i_])
            src_col = np.arange(k2)
            src_col[:nh_] = perm
            src_col[k2 - 1 - np.arange(nh_)] = k2 - 1 - perm
            one = np.ones(k2, np.int64)
            cp[i_] = False
            self.colcopy(UEp[i_] + src_col * ld1[i_] * el, Vt[i_] + np.arange(k2) * ld1[i_] * el, n[i_] * one, one,
                         ld1[i_] * one, ld1[i_] * one)
        self.colcopy(UEp[cp], Vt[cp], n[cp], 2 * ke[cp], ld1[cp], ld1[cp])
        # ---- eigenvalue-1/2 modes (pfaffian.py:807-816, :867-874, :883-889) ---------------------------
        # The Ritz vectors of the degenerate 1/2 eigenspace are an arbitrary orthonormal basis; the rest of
        # the sweep needs conjugate pairs (u, conj u) like every other mode.  The eigenspace is closed
        # under conjugation, so Re(U_h G) (G random) spans its real form; it is orthonormalised to a real
        # basis w, shuffled with the reference's fixed-seed orthogonal matrix O, and combined as
        # (w_j + i w_(kh+j)) / sqrt(2) (left) resp. (-i w_j + w_(kh+j)) / sqrt(2) (right).
        hm = np.nonzero((kh > 0) & doE)[0]
        if hm.size:
            from scipy.stats import ortho_group

            khm = kh[hm]
            mx = int(2 * khm.max())
            rng = np.random.default_rng(4321)
            t_G = self._up(np.ascontiguousarray((rng.standard_normal((mx, mx)) + 1j * rng.standard_normal((mx, mx))).T))
            oT, tT = offsets(n[hm] * 2 * khm)
            d_Th = self._alloc(tT)
            Tp = d_Th.data_ptr() + oT * el
            Uh = Vt[hm] + (ke[hm] - khm) * ld1[hm] * el          # the 2 kh contiguous 1/2 columns
            self.gemm(0, 1.0, 0.0, Uh, np.full(hm.size, t_G.data_ptr()), Tp, n[hm], 2 * khm, 2 * khm, ld1[hm],
                      np.full(hm.size, mx), ld1[hm])
            self.colcopy(Tp, Tp, n[hm], 2 * khm, ld1[hm], ld1[hm], reverse=4)      # real part, normalised
            d_scrh = self._alloc(hm.size * (mx + 1) * PANEL_W)
            self.bcgs2(Tp, n[hm], ld1[hm], np.zeros(hm.size, np.int64), 2 * khm,
                       d_scrh.data_ptr() + np.arange(hm.size) * (mx + 1) * PANEL_W * el)
            Mp = np.zeros(hm.size, np.int64)
            cache = {}
            for t_, i in enumerate(hm):
                key = (int(kh[i]), int(cs_side[i]))
                if key not in cache:
                    k_ = key[0]
                    O = ortho_group.rvs(2 * k_, random_state=1234)                 # pfaffian.py:870
                    if key[1] == 0:      # lower 1/2 column ke + j
                        M_ = (O[:, :k_] + 1j * O[:, k_:]) / 2**0.5
                    else:                # upper 1/2 column ke - 1 - j  ->  block column kh - 1 - j
                        M_ = ((-1j * O[:, :k_] + O[:, k_:]) / 2**0.5)[:, ::-1]
                    cache[key] = self._up(np.ascontiguousarray(M_.T))
                Mp[t_] = cache[key].data_ptr()
            prim = Vt[hm] + np.where(cs_side[hm] == 0, ke[hm], ke[hm] - khm) * ld1[hm] * el
            conj_ = Vt[hm] + np.where(cs_side[hm] == 0, ke[hm] - khm, ke[hm]) * ld1[hm] * el
            self.gemm(0, 1.0, 0.0, Tp, Mp, prim, n[hm], khm, 2 * khm, ld1[hm], 2 * khm, ld1[hm])
            self.colcopy(prim, conj_, n[hm], khm, ld1[hm], ld1[hm], reverse=3)      # reversed + conjugated
        kc = int(ke[centre_L])
        if kc > 0 and n[centre_R] > 0:
            # paired upper modes of the centre's right side: normalise(C_RL v_L,i) (block_svd, pfaffian.py:855)
            d_pair = self._alloc(n[centre_R] * kc)
            self.gemm(0, 1.0, 0.0, [off[centre_R]], [Vt[centre_L] + kc * ld1[centre_L] * el], [d_pair.data_ptr()],
                      [n[centre_R]], [kc], [m[centre_R]], [D], [ld1[centre_L]], [ld1[centre_R]])
            # ordered Gram-Schmidt of the partners (strongest singular value first), as in Engine.run:
            # partners of weak modes carry errors ~1e-6 from the division by sigma
            nR, ldR = int(n[centre_R]), int(ld1[centre_R])
            lamL = lam_side[centre_L][kc:]
            order = np.argsort(-(lamL * (1.0 - lamL)), kind="stable")
            Pm, Sm = np.zeros((kc, kc), np.complex128), np.zeros((kc, kc), np.complex128)
            for a_, j_ in enumerate(order):
                Pm[j_, a_] = 1.0
                Sm[a_, kc - 1 - j_] = 1.0                       # reversal only (no anticommutation signs here)
            t_P = self._up(np.ascontiguousarray(Pm.T).reshape(-1))
            t_S = self._up(np.ascontiguousarray(Sm.T).reshape(-1))
            d_T = self._alloc(nR * kc)
            self.gemm(0, 1.0, 0.0, [d_pair.data_ptr()], [t_P.data_ptr()], [d_T.data_ptr()], [nR], [kc], [kc], [ldR],
                      [kc], [ldR])
            d_scrc = self._alloc((kc + 1) * PANEL_W)
            self.bcgs2(np.array([d_T.data_ptr()]), np.array([nR]), np.array([ldR]), np.array([0]), np.array([kc]),
                       np.array([d_scrc.data_ptr()]))
            self.gemm(0, 1.0, 0.0, [d_T.data_ptr()], [t_S.data_ptr()], [Vt[centre_R]], [nR], [kc], [kc], [ldR], [kc],
                      [ldR])
            # conjugate partners: column kc + c = conj(column kc - 1 - c)
            self.colcopy([Vt[centre_R]], [Vt[centre_R] + kc * ldR * el], [nR], [kc], [ldR], [ldR], reverse=3)
        maxnb = int(nb_.max())
        if maxnb > 0:
            d_OmF = self._alloc(D * maxnb)
            nat.check(self.lib.tmf_fill_normal(self.dtype, d_OmF.data_ptr(), D * maxnb, 0xF111EE, self.stream), "fill")
            Vf = Vt + 2 * ke * ld1 * el
            self.nested_products("A", D, Cp, d_OmF.data_ptr(), D, 2 * cs_b, cs_side, Vf, nb_, ld1)  # Engine.run, stage F
            d_scr2 = self._alloc(int((ncol.max() + 1) * PANEL_W) * ncs)
            scr2 = d_scr2.data_ptr() + np.arange(ncs) * int((ncol.max() + 1) * PANEL_W) * el
            has = nb_ > 0
            self.bcgs2(Vt[has], n[has], ld1[has], 2 * ke[has], ncol[has], scr2[has], cholqr=self.filled_cholqr)   # as in Engine.run, stage F; two projection passes
        # ---- self-check of every cut side (pfaffian.py:919 -> testing.py:131-177): kept columns
        # orthonormal, and A = V diag(lambda | 1) V^H (the empty modes, conj(filled), carry eigenvalue 0)
        chk_names, d_chk = [], None
        if self.checks:
            items = []
            for i in range(ncs):
                if n[i] == 0 or ncol[i] == 0:
                    continue
                q_ = int(ncol[i])
                sdn = "L" if cs_side[i] == 0 else "R"
                items.append(dict(T=0, X=Vt[i], Y=Vt[i], w=None, rows=q_, cols=q_, q=0, inner=int(n[i]),
                                  ldx=int(ld1[i]), ldy=int(ld1[i]), mode=1))
                chk_names.append(f"v{sdn} is not unitary (cut {cs_b[i]})")
                items.append(dict(T=blk[i], X=Vt[i], Y=Vt[i], w=np.concatenate((lam_side[i], np.ones(int(nb_[i])))),
                                  rows=int(n[i]), cols=int(n[i]), q=q_, ldt=D, ldx=int(ld1[i]), ldy=int(ld1[i]),
                                  mode=0))
                chk_names.append(f"v{sdn} does not diagonalise C_{sdn}{sdn} (cut {cs_b[i]})")
            d_chk = self.recon_errors(items)
        # ---- Bogoliubov matrices v = [a | a^dag] in the complex-fermion basis (pfaffian.py:880-895) ----
        oVC, tVC = offsets(n * n)
        d_VC, d_VW = self._alloc(tVC), self._alloc(tVC)
        VC, VW = d_VC.data_ptr() + oVC * el, d_VW.data_ptr() + oVC * el
        cs_off, cs_tot = offsets(n)
        col_src, col_conj = np.zeros(cs_tot + 1, np.int32), np.zeros(cs_tot + 1, np.int8)
        for i in range(ncs):
            k_, q_, o_ = int(ke[i]), int(ns[i]), int(cs_off[i])
            if q_ == 0:
                continue
            src, cj = np.zeros(2 * q_, np.int32), np.zeros(2 * q_, np.int8)
            if cs_side[i] == 0:   # lower modes: [empty = conj(filled) | entangled lower, ascending]; upper = conj
                src[: q_ - k_] = 2 * k_ + np.arange(q_ - k_)
                cj[: q_ - k_] = 1
                src[q_ - k_: q_] = 2 * k_ - 1 - np.arange(k_)
                src[q_:], cj[q_:] = src[:q_], 1 - cj[:q_]
            else:                 # upper modes: [entangled upper, ascending | filled]; lower = conj
                src[q_: q_ + k_] = k_ - 1 - np.arange(k_)
                src[q_ + k_:] = 2 * k_ + np.arange(q_ - k_)
                src[:q_], cj[:q_] = src[q_:], 1
            col_src[o_: o_ + 2 * q_], col_conj[o_: o_ + 2 * q_] = src, cj
        t_cs, t_cj = self._up(col_src), self._up(col_conj)
        ad = np.zeros(ncs, nat.nambu_asm_desc)
        ad["src"], ad["dst"] = Vt, VC
        ad["col_src"], ad["col_conj"] = t_cs.data_ptr() + cs_off * 4, t_cj.data_ptr() + cs_off
        ad["n2"], ad["lds_"], ad["ldd"] = n, ld1, ld1
        sel = np.nonzero(n > 0)[0]
        t_ad = self._up(ad[sel])
        nat.check(self.lib.tmf_nambu_assemble_batched(t_ad.data_ptr(), len(sel), self.stream), "nambu_assemble")
        # ---- vacuum parities from det(v) (LU on a copy) ---------------------------------------------
        d_VW.copy_(d_VC)
        d_pdet = self._alloc(ncs, zero=True)
        self._lu_schur(np.asarray(VW)[sel], (d_pdet.data_ptr() + np.arange(ncs) * el)[sel], n[sel], n[sel], n[sel], np.asarray(ld1)[sel], "lu")
        self._tick("F_modes_parity", t0)

        t0 = time.perf_counter()
        h_pdet = d_pdet.cpu().numpy()
        par = np.zeros(ncs, np.int64)
        for i in range(ncs):
            if ns[i] == 0:
                continue
            sgn = -1.0 if (ns[i] * (ns[i] - 1) // 2) % 2 else 1.0
            dv = h_pdet[i].real * sgn
            if abs(abs(dv) - 1.0) > 1e-6 or abs(h_pdet[i].imag) > 1e-6:
                raise ValueError(f"cut {cs_b[i]}: Bogoliubov matrix is not unitary (det = {h_pdet[i]})")
            par[i] = 0 if dv > 0 else 1
        total_parity = int((par[centre_L] + par[centre_R]) % 2)
        pL, pR = np.zeros(L + 1, np.int64), np.zeros(L + 1, np.int64)
        for b in range(L + 1):
            if b <= oc:
                pL[b] = par[cidx[(b, 0)]]
                pR[b] = par[centre_R] if b == oc else (total_parity + pL[b]) % 2   # pfaffian.py:902-903
            else:
                pR[b] = par[cidx[(b, 1)]]
                pL[b] = (total_parity + pR[b]) % 2                                  # pfaffian.py:909-910
        flip_centre_R = pL[oc] == 1                                                 # pfaffian.py:915-916

        # ---- enumeration (same native routine as the Slater path, no filled offsets) ------------------
        sectors = _sector_list(trunc, L)
        bonds = []
        for b in range(L + 1):
            e_b = e_cut[cidx[(b, 0)]] if b <= oc else e_cut[cidx[(b, 1)]]
            sets_m, lam_raw, q, _ = nat.cut_vectors(e_b, 0, trunc.chi_max or 0, trunc.svd_min, trunc.degeneracy_tol,
                                                    sectors)
            if len(lam_raw) == 0:
                raise ValueError("No Schmidt vectors left after filtering by `trunc_par.sectors`!")
            kk = len(e_b)
            sb_ = np.zeros((len(sets_m), kk), bool)
            for i in range(kk):
                sb_[:, i] = (sets_m[:, i // 64] >> np.uint64(i % 64)) & np.uint64(1)
            idx = parity_n_argsort(sb_.sum(axis=1))                                   # pfaffian.py:1195-1197
            bonds.append(PfBond(b, e_b, int(pL[b]), int(pR[b]), sb_[idx], lam_raw[idx]))
        self._tick("host_enumerate", t0)

        # ---- per-site integer preparation (pfaffian.py:1617-1733, :1361-1408, :1766-1776) ---------------
        t0 = time.perf_counter()
        prep = []
        for i in range(L):
            mode = 0 if i < oc else 1
            side = "L" if mode == 0 else "R"
            bb, kb = (i, i + 1) if mode == 0 else (i + 1, i)
            B, K = bonds[bb], bonds[kb]
            ib, ik = cidx[(bb, mode)], cidx[(kb, mode)]
            nbs, nks = int(ns[ib]), int(ns[ik])          # sites: nks = nbs + 1
            sets1 = B.side_sets(side)
            chi_b = len(sets1)
            z, o = np.zeros((chi_b, 1), bool), np.ones((chi_b, 1), bool)
            sets1 = np.block([[sets1, z], [sets1, o]]) if mode == 0 else np.block([[z, sets1], [o, sets1]])
            fix = B.parity(side) % 2 != K.parity(side) % 2
            u_p = -1 if (mode == 0 and B.parity("L") % 2 == 1) else 1
            # rows of Vr = columns of the extended bra matrix: [a (nks) | a^dag (nks)]
            rs, rg = np.zeros(2 * nks, np.int32), np.ones(2 * nks, np.int8)
            if mode == 0:   # physical orbital last (pfaffian.py:1667-1673); its rows are -1 (c^dag) and -2 (c)
                rs[:nbs], rs[nbs] = np.arange(nbs), -1
                rs[nks: nks + nbs], rs[nks + nbs] = nbs + np.arange(nbs), -2
                rg[nbs] = rg[nks + nbs] = u_p
                if fix:     # swap the physical a and a^dag columns, flip the last set bit (pfaffian.py:1711-1713)
                    rs[nbs], rs[nks + nbs] = -2, -1
                    sets1 = sets1.copy()
                    sets1[:, -1] = ~sets1[:, -1]
            else:           # physical orbital first (pfaffian.py:1682-1688)
                rs[0], rs[1: nks] = -1, np.arange(nbs)
                rs[nks], rs[nks + 1:] = -2, nbs + np.arange(nbs)
                if fix:     # pfaffian.py:1714-1719: all Bogoliubov columns negated, physical pair swapped
                    rg[:] = -1
                    rs[0], rs[nks] = -2, -1
                    rg[0] = rg[nks] = 1
                    sets1 = sets1.copy()
                    sets1[:, 0] = ~sets1[:, 0]
            sets2 = K.side_sets(side)
            act1, act2 = sets1.shape[1], sets2.shape[1]
            i1 = np.nonzero(sets1.any(axis=0))[0]
            i2 = np.nonzero(sets2.any(axis=0))[0][::-1]
            s1, s2 = sets1[:, i1], sets2[:, i2]
            if mode == 0:   # active modes at the end (pfaffian.py:1377-1379)
                i1, i2 = i1 + nks - act1, i2 + nks - act2
            na_, nb2 = len(i1), len(i2)
            new1 = np.concatenate((np.zeros((len(s1), nb2), bool), s1), axis=1)
            new2 = np.concatenate((s2, np.zeros((len(s2), na_), bool)), axis=1)
            leg_idx = parity_n_argsort(new1.sum(axis=1))
            new1 = new1[leg_idx]
            idx_n_bra = _bunched(new1.sum(axis=1))
            blocks = []
            for nbq, (r0, r1) in idx_n_bra.items():
                for nkq, (c0, c1) in K.idx_n.items():
                    if (nbq + nkq) % 2 == 0:
                        blocks.append((nbq, nkq, r0, r1, c0, c1))
            prep.append(dict(mode=mode, ib=ib, ik=ik, nbs=nbs, nks=nks, rs=rs, rg=rg, i1=i1.astype(np.int32),
                             i2=i2.astype(np.int32), na=na_, nb=nb2, new1=new1, new2=new2, leg_idx=leg_idx,
                             blocks=blocks, chi_b=chi_b, chi_k=len(sets2),
                             col_sign=-1 if (mode == 1 and kb == oc and flip_centre_R) else 1))
        self._tick("host_site_prepare", t0)

        # ---- device: Vr = ext^H v_ket, W, LU/Schur, Pfaffian matrix -------------------------------------
        t0 = time.perf_counter()
        ib = np.array([r["ib"] for r in prep])
        ik = np.array([r["ik"] for r in prep])
        mode = np.array([r["mode"] for r in prep])
        nb2n, nk2n = n[ib], n[ik]                       # 2 nbs, 2 nks
        Lk = nk2n // 2
        na_v = np.array([r["na"] for r in prep])
        nb_v = np.array([r["nb"] for r in prep])
        oO, tO = offsets(nb2n * nk2n)
        oVr, tVr = offsets(nk2n * nk2n)
        mw = Lk + na_v + nb_v
        oW, tW = offsets(mw * mw)
        mn = na_v + nb_v
        oN, tN = offsets(mn * mn)
        d_O, d_Vr, d_W, d_N = self._alloc(tO), self._alloc(tVr), self._alloc(tW), self._alloc(tN)
        d_det, d_norm = self._alloc(L), self._alloc(L)
        Op, Vrp, Wp, Np = (t.data_ptr() + o * el for t, o in ((d_O, oO), (d_Vr, oVr), (d_W, oW), (d_N, oN)))
        detp, normp = d_det.data_ptr() + np.arange(L) * el, d_norm.data_ptr() + np.arange(L) * el
        Vk_sub = VC[ik] + np.where(mode == 1, 2, 0) * el     # right mode: the site's rows are rows 0, 1 of the ket
        physp = VC[ik] + np.where(mode == 1, 0, nb2n) * el
        self.gemm(1, 1.0, 0.0, VC[ib], Vk_sub, Op, nb2n, nk2n, nb2n, ld1[ib], ld1[ik], np.maximum(nb2n, 1))
        rs_off, rs_tot = offsets(nk2n)
        row_sel = np.concatenate([r["rs"] for r in prep]).astype(np.int32)
        row_sign = np.concatenate([r["rg"] for r in prep]).astype(np.int8)
        col_sel = np.concatenate([np.arange(x, dtype=np.int32) for x in nk2n])
        col_sign = np.concatenate([np.full(x, r["col_sign"], np.int8) for x, r in zip(nk2n, prep)])
        t_rs, t_rg, t_cs2, t_cg = self._up(row_sel), self._up(row_sign), self._up(col_sel), self._up(col_sign)
        gd = np.zeros(L, nat.gather_desc)
        gd["src"], gd["dst"] = Op, Vrp
        gd["row_sel"], gd["col_sel"] = t_rs.data_ptr() + rs_off * 4, t_cs2.data_ptr() + rs_off * 4
        gd["row_sign"], gd["col_sign"] = t_rg.data_ptr() + rs_off, t_cg.data_ptr() + rs_off
        gd["phys"] = physp
        gd["rows"], gd["cols"] = nk2n, nk2n
        gd["lds_"], gd["ldd"], gd["ldp"] = np.maximum(nb2n, 1), nk2n, ld1[ik]
        t_gd = self._up(gd)
        nat.check(self.lib.tmf_gather_signed_batched(self.dtype, t_gd.data_ptr(), L, self.stream), "gather")
        i1_off, i1_tot = offsets(na_v)
        i2_off, i2_tot = offsets(nb_v)
        t_i1 = self._up(np.concatenate([r["i1"] for r in prep] + [np.zeros(1, np.int32)]))
        t_i2 = self._up(np.concatenate([r["i2"] for r in prep] + [np.zeros(1, np.int32)]))
        wd = np.zeros(L, nat.nambu_w_desc)
        wd["Vr"], wd["W"] = Vrp, Wp
        wd["idx1"], wd["idx2"] = t_i1.data_ptr() + i1_off * 4, t_i2.data_ptr() + i2_off * 4
        wd["L"], wd["na"], wd["nb"], wd["ldv"], wd["ldw"] = Lk, na_v, nb_v, nk2n, mw
        t_wd = self._up(wd)
        nat.check(self.lib.tmf_nambu_w_batched(t_wd.data_ptr(), L, self.stream), "nambu_w")
        self._lu_schur(Wp, detp, mw, mw, Lk, mw, "lu_schur")
        pd = np.zeros(L, nat.pf_matrix_desc)
        pd["S"], pd["N"] = Wp + (Lk + Lk * mw) * el, Np
        pd["na"], pd["nb"], pd["lds_"], pd["ldn"] = na_v, nb_v, mw, np.maximum(mn, 1)
        t_pd = self._up(pd)
        nat.check(self.lib.tmf_pf_matrix_batched(t_pd.data_ptr(), L, self.stream), "pf_matrix")
        # Onishi norm sqrt(prod sv(U)) = |det U|^(1/2) (pfaffian.py:1352-1359): tiny, on the host
        nat.check(self.lib.tmf_onishi_norms(d_det.data_ptr(), d_norm.data_ptr(), L, self.stream), "tmf_onishi_norms")
        self._tick("S_overlap_schur", t0)

        # ---- all sub-Pfaffians ------------------------------------------------------------------------------
        t0 = time.perf_counter()
        out_sizes = [sum((r1 - r0) * (c1 - c0) for (_, _, r0, r1, c0, c1) in r["blocks"]) for r in prep]
        out_off, out_tot = offsets(np.array(out_sizes))
        d_out = self._alloc(out_tot)
        a16 = lambda x: (x + 15) & ~15  # noqa: E731
        # Tile descriptors of every block of every site, vectorised over the ~12 000 blocks of a conversion (the loop over
        # blocks with two nonzero() calls each was 120 ms of host time at config 4).  Index pool per site: the occupied-
        # orbital lists of all bra rows (row-major nonzero), then those of all ket rows; the rows of a block are contiguous,
        # so its lists are slices of them.
        pool_parts, spo, len1, off1_parts, off2_parts, rb1, rb2 = [], [], [], [], [], [], []
        pool_off = ro1 = ro2 = 0
        for r in prep:
            cols1 = np.nonzero(r["new1"])[1].astype(np.uint8)
            cols2 = np.nonzero(r["new2"])[1].astype(np.uint8)
            o1 = np.concatenate(([0], np.cumsum(r["new1"].sum(axis=1)))).astype(np.int64)
            o2 = np.concatenate(([0], np.cumsum(r["new2"].sum(axis=1)))).astype(np.int64)
            pool_parts += [cols1, cols2]
            spo.append(pool_off), len1.append(cols1.size)
            pool_off += cols1.size + cols2.size
            off1_parts.append(o1), off2_parts.append(o2)
            rb1.append(ro1), rb2.append(ro2)
            ro1 += o1.size
            ro2 += o2.size
        pool = np.concatenate(pool_parts + [np.zeros(1, np.uint8)]) if pool_parts else np.zeros(1, np.uint8)
        t_pool = self._up(pool)
        nblk = np.array([len(r["blocks"]) for r in prep], np.int64)
        blk = np.array([b for r in prep for b in r["blocks"]], np.int64).reshape(-1, 6)
        site = np.repeat(np.arange(len(prep)), nblk)
        nbq, nkq, r0, r1, c0, c1 = (blk[:, j] for j in range(6))
        nsb, nsk, mq = r1 - r0, c1 - c0, nbq + nkq
        if blk.shape[0] and int(mq.max()) > 32:
            raise NotImplementedError(f"sub-Pfaffian of order {int(mq.max())} > 32")
        spo, len1, rb1, rb2 = (np.array(x, np.int64) for x in (spo, len1, rb1, rb2))
        off1_cat = np.concatenate(off1_parts) if off1_parts else np.zeros(1, np.int64)
        off2_cat = np.concatenate(off2_parts) if off2_parts else np.zeros(1, np.int64)
        bo = spo[site] + off1_cat[rb1[site] + r0]
        ko = spo[site] + len1[site] + off2_cat[rb2[site] + c0]
        assert np.all(off1_cat[rb1[site] + r1] - off1_cat[rb1[site] + r0] == nsb * nbq)
        assert np.all(off2_cat[rb2[site] + c1] - off2_cat[rb2[site] + c0] == nsk * nkq)
        o_blk = np.concatenate(([0], np.cumsum(nsb * nsk)))[:-1]              # blocks are listed site by site: = out_off[site] + ...
        ta = np.maximum(1, np.minimum(nsb, -(-2048 // np.maximum(nsk, 1))))
        G = np.where(mq <= 8, 8, np.where(mq <= 16, 16, 32))
        mn_b = np.asarray(mn, np.int64)[site]
        need = a16(mn_b ** 2 * el) + a16(nsk * nkq) + a16(ta * nbq) + (256 // G) * 2 * np.maximum(mq, 1) * el + 16
        ntile = -(-nsb // ta)
        rep = np.repeat(np.arange(blk.shape[0]), ntile)
        first = np.concatenate(([0], np.cumsum(ntile)))[:-1]
        a0 = (np.arange(rep.size) - first[rep]) * ta[rep]
        a1 = np.minimum(nsb[rep], a0 + ta[rep])
        Np_a, normp_a = np.asarray(Np, np.uint64), np.asarray(normp, np.uint64)
        split = np.cumsum(nblk)[:-1]
        for r, ob in zip(prep, np.split(o_blk, split)):
            r["block_out"] = [int(x) for x in ob]
        for m_ in np.unique(mq):
            sel = rep[mq[rep] == m_]
            w = mq[rep] == m_
            dd = np.zeros(int(w.sum()), nat.pf_desc)
            dd["N"], dd["scale"] = Np_a[site[sel]], normp_a[site[sel]]
            dd["bra_idx"] = t_pool.data_ptr() + bo[sel]
            dd["ket_idx"] = t_pool.data_ptr() + ko[sel]
            dd["out"] = d_out.data_ptr() + o_blk[sel] * el
            dd["nn"], dd["ldn"] = mn_b[sel], np.maximum(mn_b[sel], 1)
            dd["n1"], dd["n2"], dd["nsb"], dd["nsk"] = nbq[sel], nkq[sel], nsb[sel], nsk[sel]
            dd["a0"], dd["a1"] = a0[w], a1[w]
            t_dd = self._up(dd)
            nat.check(self.lib.tmf_pf_gather_batched(self.dtype, int(m_), t_dd.data_ptr(), len(dd), int(need[mq == m_].max()),
                                                     self.stream), "tmf_pf_gather_batched")
        self._tick("S_pfaffians", t0)

        t0 = time.perf_counter()
        h_out = d_out.cpu().numpy()
        norms = d_norm.cpu().numpy()
        sites = []
        for i, r in enumerate(prep):
            blocks = {}
            for (nbq, nkq, r0, r1, c0, c1), o in zip(r["blocks"], r["block_out"]):
                blocks[(nbq, nkq)] = (r0, r1, c0, c1, h_out[o: o + (r1 - r0) * (c1 - c0)].reshape(r1 - r0, c1 - c0))
            bb, kb = (i, i + 1) if r["mode"] == 0 else (i + 1, i)
            qtotal = 0 if r["mode"] == 0 else int((bonds[bb].parity() + bonds[kb].parity()) % 2)
            sites.append(PfSite("left" if r["mode"] == 0 else "right", float(norms[i].real), qtotal, r["leg_idx"],
                                blocks, r["chi_b"], r["chi_k"]))
        self._tick("download", t0)
        self.check_results = {}
        if d_chk is not None:
            h_chk = d_chk.cpu().numpy()
            # one entry per message: the worst cut
            for nm, v in zip(chk_names, h_chk):
                key = nm.split(" (cut")[0]
                if not v <= self.check_results.get(key, (-1.0, ""))[0]:
                    self.check_results[key] = (float(v), nm)
            self.check_results = {nm: v for v, nm in self.check_results.values()}
        self.timings["total"] = time.perf_counter() - t_all
        self._keep.clear()
        return self._finish(PfMPSData(bonds, sites, oc, unit_cell_width, dict(self.timings)))
