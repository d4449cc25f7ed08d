"""Finite -> infinite MPS conversion (temfpy/iMPS.py) on MI355X.

``MPS_to_iMPS`` (iMPS.py:232-441), ``overlap_schmidt`` (:20-62), ``basis_rotation`` (:65-192) and ``iMPSError``
(:195-229) with the reference's names, arguments, defaults, warnings and exceptions, on the finite MPS objects
of this package (``MPSData`` from ``slater.C_to_MPS``: U(1) charge blocks; ``PfMPSData`` from
``pfaffian.C_to_MPS``: parity blocks, indices of every bond reordered by parity; ``SpinMPSData`` from the Gutzwiller
projections: 2 S^z blocks - the projected iMPS is obtained from the projected finite chains of two lengths, the
reference's own route of projecting an infinite MPS, ``canonical_form_infinite1``, is not built).

Device work (C ABI, ``include/temfpy_hip.h``; no CPU path for it):

* the Schmidt-vector overlaps ``<L'_a|L_b>`` / ``<R'_b|R_a>`` of the two chains (TeNPy ``TransferMatrix.matvec``
  at iMPS.py:46-60): one transfer-matrix step per site = three batched MFMA launches over all charge sectors
  (``E K^p``, then ``(K'^p)^H (E K^p)`` accumulated over p) with the tensor blocks regrouped once by
  ``tmf_copy_blocks_batched``; canonical-form changes ``A <-> B`` (TeNPy ``get_B``) are diagonal GEMMs;
* the orthogonal-Procrustes rotations (``npc.svd`` at iMPS.py:166 / :170, ``U V`` at :171): one-sided Jacobi with
  accumulated rotations (``tmf_jacobi_compact_batched``) on the Schmidt-weighted overlap of every charge sector,
  ``U`` = normalised columns of ``M V``, rotation = ``U V^H`` by the GEMM kernel;
* the gauge fixing of the unit cell (``tensordot(C, B_0)``, ``tensordot(B_last, D)``, iMPS.py:420-421).

The scalar diagnostics (unitarity and Schmidt-mixing errors, iMPS.py:137, :183) are evaluated on the host from
the (chi x chi) overlap and rotation matrices that are returned anyway.
"""
from __future__ import annotations

import logging
import warnings
from typing import Iterable, Literal, NamedTuple

import numpy as np

from . import _native as nat
from .gutzwiller import _as_fermions, _gemm_recs, _gemm_tiles, _sector_table, _cdiv
from .testing import assert_array_less

logger = logging.getLogger(__name__)

_NUMERICAL_TOL = 1e-14
_UNITARY_TOL = 1e-6
_SCHMIDT_TOL = 1e-6


class iMPSError(NamedTuple):
    """Container of the approximation errors accrued by :func:`MPS_to_iMPS` (iMPS.py:195-229)."""

    left_unitary: float
    left_schmidt: float
    right_unitary: float
    right_schmidt: float

    @property
    def left_total(self) -> float:
        return (self.left_schmidt**2 + self.left_unitary**2) ** 0.5

    @property
    def right_total(self) -> float:
        return (self.right_schmidt**2 + self.right_unitary**2) ** 0.5

    @property
    def total_error(self) -> float:
        return float(np.linalg.norm(self))

    def __repr__(self) -> str:
        fields = [f"    {f}={x:.8e}" for f, x in zip(self._fields, self) if x != 0]
        if len(fields) == 0:
            return "iMPSError()"
        return "iMPSError(\n" + (",\n".join(fields)) + "\n)"


class BlockMatrix:
    """Charge-block-diagonal matrix: ``blocks[(q_row, q_col)]`` = ndarray; ``rows`` / ``cols``: per-index charges.
    Stands in for the ``npc.Array`` with legs ``vL`` (rows) and ``vR`` (columns) of iMPS.py."""

    def __init__(self, blocks, rows, cols):
        self.blocks, self.rows, self.cols = blocks, np.asarray(rows), np.asarray(cols)

    def dense(self):
        rt, ct = _sector_table(self.rows), _sector_table(self.cols)
        dt = np.result_type(*[b.dtype for b in self.blocks.values()], float) if self.blocks else float
        M = np.zeros((len(self.rows), len(self.cols)), dt)
        for (qr, qc), b in self.blocks.items():
            M[rt[qr][0]: rt[qr][0] + rt[qr][1], ct[qc][0]: ct[qc][0] + ct[qc][1]] = b
        return M


class iMPSData:
    """Infinite MPS unit cell in right-canonical form: ``blocks[i]`` = list of (p, q_l, q_r, l0, l1, r0, r1, array),
    ``lam`` (L + 1 entries, first = last), per-bond ``charges`` (offset subtracted), ``cell_charge`` = particles
    per unit cell (charge rule q_l + p = q_r, except on the last site where q_l + p = q_r + cell_charge)."""

    bc = "infinite"

    def __init__(self, blocks, lam, charges, cell_charge, unit_cell_width, conserve=None):
        self.blocks, self.lam, self.charges = blocks, lam, charges
        self.L = len(blocks)
        self.cell_charge, self.unit_cell_width = cell_charge, unit_cell_width
        self.conserve = conserve      # 'N' / 'parity' (FermionSite) or 'spin Sz' / 'spin None' (SpinHalfSite) of the two chains
        self.form = ["B"] * self.L

    @property
    def chi(self):
        return [len(x) for x in self.lam]

    def dense_tensors(self):
        out = []
        for j, bl in enumerate(self.blocks):
            dt = np.result_type(*[b[7].dtype for b in bl], float) if bl else float
            T = np.zeros((2, len(self.lam[j]), len(self.lam[j + 1])), dt)
            for p, _, _, l0, l1, r0, r1, a in bl:
                T[p, l0:l1, r0:r1] = a
            out.append(T)
        return out

    def to_tenpy(self):
        """``tenpy.networks.mps.MPS(..., bc="infinite", form="B")`` of the unit cell, as slater.py:1555-1562 /
        pfaffian.py:2083-2090 build it.  Needs physics-tenpy, which is not installed in the build environment: written
        against its documented interface and exercised by no test here; the assembled tensors are read back and compared
        with the cell's own, a mismatch raises."""
        from tenpy import networks

        if self.conserve in ("N", "parity"):
            site = networks.site.FermionSite(self.conserve)
        elif self.conserve in ("spin Sz", "spin None"):
            site = networks.site.SpinHalfSite("Sz" if self.conserve == "spin Sz" else None)
        else:
            raise ValueError(f"unknown charge kind {self.conserve!r}")
        return cell_to_tenpy(self, site, charged=self.conserve != "spin None")


def cell_to_tenpy(cell, site, charged=True):
    """Shared by ``iMPSData.to_tenpy`` and ``gutzwiller.SpiniMPSData.to_tenpy``: B tensors with legs (vL, p, vR), the charges
    of bond L being those of bond 0 (the last tensor then carries the cell's charge as its ``qtotal``, which
    ``npc.Array.from_ndarray`` detects)."""
    import tenpy.linalg.np_conserved as npc
    from tenpy import networks

    chinfo = site.leg.chinfo
    Bs = []
    for j, T in enumerate(cell.dense_tensors()):
        if charged:
            legs = [npc.LegCharge.from_qflat(chinfo, [[int(q)] for q in cell.charges[j]], qconj=+1), site.leg,
                    npc.LegCharge.from_qflat(chinfo, [[int(q)] for q in cell.charges[(j + 1) % cell.L]], qconj=-1)]
            B = npc.Array.from_ndarray(T.transpose(1, 0, 2), legs, labels=["vL", "p", "vR"], cutoff=0.0)
        else:
            B = npc.Array.from_ndarray_trivial(T.transpose(1, 0, 2), labels=["vL", "p", "vR"])
        if not np.array_equal(B.to_ndarray(), T.transpose(1, 0, 2)):
            raise RuntimeError(f"TeNPy assembly self-check failed at site {j} of the unit cell")
        Bs.append(B)
    psi = networks.mps.MPS([site] * cell.L, Bs, [np.asarray(x) for x in cell.lam], bc="infinite", form="B",
                           unit_cell_width=cell.unit_cell_width)
    psi._temfpy_amd = cell
    return psi


# ---------------------------------------------------------------------------------------------------
# device helper
# ---------------------------------------------------------------------------------------------------
class _Dev:
    def __init__(self, device, cplx):
        import torch

        if not torch.cuda.is_available():
            raise nat.NativeError("temfpy_amd.iMPS needs a HIP device; there is no CPU fallback")
        self.torch, self.device, self.lib = torch, torch.device(device), nat.load()
        self.cplx = cplx
        self.dt = nat.TMF_C128 if cplx else nat.TMF_F64
        self.np_dt = np.dtype(np.complex128 if cplx else np.float64)
        self.el = self.np_dt.itemsize
        self.keep = []

    @property
    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def zeros(self, n):
        t = self.torch.zeros(max(int(n), 1), dtype=self.torch.complex128 if self.cplx else self.torch.float64,
                             device=self.device)
        self.keep.append(t)
        return t

    def up(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.keep.append(t)
        return t

    def gemm(self, items, opA=0, alpha=1.0, beta=0.0):
        if not items:
            return
        d = _gemm_recs(items)
        tiles, tn = _gemm_tiles(d)
        td, tt = self.up(d.view(np.uint8).reshape(-1)), self.up(tiles.reshape(-1))
        nat.check(self.lib.tmf_gemm_batched(self.dt, opA, float(alpha), float(beta), td.data_ptr(), tt.data_ptr(), len(tiles),
                                            tn, self.stream), "tmf_gemm_batched")

    def copy(self, recs):
        if len(recs) == 0:
            return
        recs = np.array(recs, nat.copy_desc)
        t = self.up(recs.view(np.uint8).reshape(-1))
        mt = int((_cdiv(recs["rows"].astype(np.int64), 32) * _cdiv(recs["cols"].astype(np.int64), 32)).max())
        nat.check(self.lib.tmf_copy_blocks_batched(self.dt, t.data_ptr(), len(recs), mt, self.stream), "tmf_copy_blocks_batched")


class _Chain:
    """Charge blocks of one finite MPS on the device, every block an (n_l x n_r) column-major matrix."""

    def __init__(self, dev, mps):
        f = _as_fermions(mps)            # MPSData: U(1) blocks; PfMPSData: parities read off the tensors
        self.f, self.dev, self.L = f, dev, f.L
        self.mod = 2 if f.conserve == "parity" else None
        self.form = list(mps.form)
        self.lam = [np.asarray(x) if f.perm is None else np.asarray(x)[f.perm[b]] for b, x in enumerate(mps.lam)]
        self.tabs = [_sector_table(q) for q in f.charges]
        flat = f.flat.numpy() if not isinstance(f.flat, np.ndarray) else f.flat
        if flat.dtype != dev.np_dt:
            flat = flat.astype(dev.np_dt)
        d_flat = dev.up(flat)
        sizes = np.where(f.blocks["trans"] == 1, f.blocks["cols"].astype(np.int64) * f.blocks["rows"],
                         f.blocks["rows"].astype(np.int64) * f.blocks["cols"])
        off = np.concatenate(([0], np.cumsum((sizes + 1) & ~1)))
        self.arena = dev.zeros(off[-1] + 2)
        base, el = self.arena.data_ptr(), dev.el
        recs, self.blk = [], {}
        for k, r in enumerate(f.blocks):
            n_l, n_r = (int(r["cols"]), int(r["rows"])) if r["trans"] else (int(r["rows"]), int(r["cols"]))
            ptr = base + el * int(off[k])
            recs.append((d_flat.data_ptr() + el * int(r["off"]), ptr, int(r["rows"]), int(r["cols"]), int(r["ld"]), n_l,
                         1 if r["trans"] else 0, 0))
            self.blk[(int(r["site"]), int(r["p"]), int(r["cl"]))] = [ptr, n_l, n_r, int(r["cr"])]
        dev.copy(recs)

    def to_form(self, i, want):
        """Blocks of site i in canonical form `want` (TeNPy get_B): diag(lam_l^-+1) T diag(lam_r^+-1) as two
        diagonal GEMMs; returns {(p, cl): [ptr, n_l, n_r, cr]}."""
        mine = {k[1:]: v for k, v in self.blk.items() if k[0] == i}
        if self.form[i] == want:
            return mine
        dev = self.dev
        sl, sr = (1.0 / self.lam[i], self.lam[i + 1]) if want == "B" else (self.lam[i], 1.0 / self.lam[i + 1])
        dl = {c: np.diag(sl[a: a + n]).astype(dev.np_dt) for c, (a, n) in self.tabs[i].items()}
        dr = {c: np.diag(sr[a: a + n]).astype(dev.np_dt) for c, (a, n) in self.tabs[i + 1].items()}
        pl = {c: dev.up(m.reshape(-1)).data_ptr() for c, m in dl.items()}
        pr = {c: dev.up(m.reshape(-1)).data_ptr() for c, m in dr.items()}
        g1, g2, out = [], [], {}
        for (p, cl), (ptr, n_l, n_r, cr) in mine.items():
            t1, t2 = dev.zeros(n_l * n_r), dev.zeros(n_l * n_r)
            g1.append((pl[cl], ptr, t1.data_ptr(), n_l, n_r, n_l, n_l, n_l, n_l))
            g2.append((t1.data_ptr(), pr[cr], t2.data_ptr(), n_l, n_r, n_r, n_l, n_r, n_l))
            out[(p, cl)] = [t2.data_ptr(), n_l, n_r, cr]
        dev.gemm(g1)
        dev.gemm(g2)
        return out


def _transfer(dev, bra_sites, ket_sites, mode, dq, mod=None):
    """Overlap of the Schmidt vectors spanned by two tensor chains (lists of per-site block dicts); charges of
    the ket chain are those of the bra chain + dq.  Returns {bra charge: (tensor, rows, cols)} at the far end."""
    left = mode == "left"
    order = range(len(bra_sites)) if left else range(len(bra_sites) - 1, -1, -1)
    one = dev.zeros(1)
    one.fill_(1.0)
    E = None
    for i in order:
        B, K = bra_sites[i], ket_sites[i]
        if E is None:      # trivial end: a single 1 x 1 sector
            ends = {(cl if left else v[3]) for (p, cl), v in B.items()}
            assert len(ends) == 1, "the end of the chain must carry a single state"
            E = {next(iter(ends)): (one, 1, 1)}
        terms = {}
        for (p, cl), (bptr, bl, br, bcr) in B.items():
            kk = K.get((p, (cl + dq) % mod if mod else cl + dq))
            if kk is None:
                continue
            kptr, kl, kr, kcr = kk
            src, dst = (cl, bcr) if left else (bcr, cl)
            if src not in E:
                continue
            terms.setdefault(dst, []).append((p, src, bptr, bl, br, kptr, kl, kr))
        g1, first, second, newE = [], [], [], {}
        for dst, tl in terms.items():
            for n_t, (p, src, bptr, bl, br, kptr, kl, kr) in enumerate(tl):
                Et, er, ec = E[src]
                if left:      # T = E K (n'_src x n_dst), E' += B^H T (n'_dst x n_dst)
                    T = dev.zeros(er * kr)
                    g1.append((Et.data_ptr(), kptr, T.data_ptr(), er, kr, ec, er, kl, er))
                    rows, cols = br, kr
                    item = (bptr, T.data_ptr(), None, br, kr, bl, bl, er, br)
                else:         # T = K E (n_dst x n'_src), E' += T B^H (n_dst x n'_dst); B^H prepared by the caller
                    T = dev.zeros(kl * ec)
                    g1.append((kptr, Et.data_ptr(), T.data_ptr(), kl, ec, kr, kl, er, kl))
                    rows, cols = kl, bl
                    item = (T.data_ptr(), bptr, None, kl, bl, ec, kl, br, kl)
                if dst not in newE:
                    newE[dst] = (dev.zeros(rows * cols), rows, cols)
                item = item[:2] + (newE[dst][0].data_ptr(),) + item[3:]
                (first if n_t == 0 else second).append(item)
        dev.gemm(g1)
        dev.gemm(first, opA=1 if left else 0)
        dev.gemm(second, opA=1 if left else 0, beta=1.0)
        E = newE
    return E


def _conj_transposed(dev, sites):
    """B^H of every block (for the rightward contraction the bra enters as T B^H): [ptr, n_r, n_l, cr] with the
    SAME keys; the stored matrix is (n_r x n_l)."""
    out, recs = [], []
    for S in sites:
        o = {}
        for key, (ptr, n_l, n_r, cr) in S.items():
            t = dev.zeros(n_l * n_r)
            recs.append((ptr, t.data_ptr(), n_l, n_r, n_l, n_r, 3 if dev.cplx else 1, 0))
            o[key] = [t.data_ptr(), n_l, n_r, cr]
        out.append(o)
    dev.copy(recs)
    return out


def overlap_schmidt(bra, ket, mode: str, *, segment_bra=None, segment_ket=None, device: str = "cuda:0") -> BlockMatrix:
    """Overlap between two sets of Schmidt vectors (iMPS.py:20-62): ``<L'_a|L_b>`` (mode "left", rows = bra) or
    ``<R'_b|R_a>`` (mode "right", rows = ket).  The reference receives segments cut out with TeNPy's
    ``extract_segment`` (iMPS.py:382-383, :399-400); here the site ranges ``(first, last + 1)`` of the two
    ``MPSData`` are passed as ``segment_bra`` / ``segment_ket`` (default: the whole chains).  A segment must
    start (mode "left") or end (mode "right") at an end of its chain."""
    mode = mode.lower()
    if mode not in ("left", "right"):
        raise ValueError("`mode` must be either 'left' or 'right', got " + repr(mode))
    b0, b1 = segment_bra if segment_bra is not None else (0, bra.L)
    k0, k1 = segment_ket if segment_ket is not None else (0, ket.L)
    assert b1 - b0 == k1 - k0, "The two MPS have different lengths."
    cplx = any(np.iscomplexobj(x.sites[0].dense()) for x in (bra, ket))
    dev = _Dev(device, cplx)
    cb, ck = _Chain(dev, bra), _Chain(dev, ket)
    return _overlap(dev, cb, b0, b1, ck, k0, k1, mode)


def _overlap(dev, cb, b0, b1, ck, k0, k1, mode):
    left = mode == "left"
    want = "A" if left else "B"
    Bs = [cb.to_form(i, want) for i in range(b0, b1)]
    Ks = [ck.to_form(i, want) for i in range(k0, k1)]
    qb, qk = (cb.f.charges[b0], ck.f.charges[k0]) if left else (cb.f.charges[b1], ck.f.charges[k1])
    mod = cb.mod
    assert ck.mod == mod, "Incompatible ChargeInfo in the two MPS"
    dq = int(qk[0]) - int(qb[0])        # both ends carry a single state; constant along the segment
    dq = dq % mod if mod else dq
    E = _transfer(dev, _conj_transposed(dev, Bs) if not left else Bs, Ks, mode, dq, mod)
    dev.torch.cuda.synchronize(dev.device)
    end_b, end_k = (b1, k1) if left else (b0, k0)
    blocks = {}
    for c, (t, r, cc) in E.items():
        m = t.cpu().numpy()[: r * cc].reshape(cc, r).T
        ck_ = (c + dq) % mod if mod else c + dq
        blocks[(c, ck_) if left else (ck_, c)] = m
    rows, cols = (cb.f.charges[end_b], ck.f.charges[end_k]) if left else (ck.f.charges[end_k], cb.f.charges[end_b])
    return BlockMatrix(blocks, rows, cols)


def basis_rotation(overlap: BlockMatrix, Schmidt_bra: np.ndarray, Schmidt_ket: np.ndarray, mode: str, *,
                   form: str = "B", numerical_tol: float = _NUMERICAL_TOL, unitary_tol: float = _UNITARY_TOL,
                   schmidt_tol: float = _SCHMIDT_TOL, device: str = "cuda:0"):
    """Optimal unitary basis rotation between two sets of Schmidt vectors (iMPS.py:65-192): returns
    (rotation_matrix, unitary_error, schmidt_error)."""
    mode = mode.lower()
    assert mode in ["left", "right"], f"`mode` must be either 'left' or 'right', got {mode!r}"
    form = form.upper()
    assert form in ["A", "B"], f"`form` must be either 'A' or 'B', got {form!r}"
    left = mode == "left"
    S_bra, S_ket = np.asarray(Schmidt_bra, float), np.asarray(Schmidt_ket, float)
    rt, ct = _sector_table(overlap.rows), _sector_table(overlap.cols)
    s_rows, s_cols = (S_bra, S_ket) if left else (S_ket, S_bra)          # rows: bra (left) / ket (right)
    bra_side = (mode, form) in [("left", "A"), ("right", "B")]             # iMPS.py:163
    C_Sk, Ms = {}, {}
    for key, blk in overlap.blocks.items():
        sr = s_rows[rt[key[0]][0]: rt[key[0]][0] + rt[key[0]][1]]
        sc = s_cols[ct[key[1]][0]: ct[key[1]][0] + ct[key[1]][1]]
        C_Sk[key] = blk * sc[None, :] if left else sr[:, None] * blk       # overlap.scale_axis(Schmidt_ket, v_ket), :135
        if bra_side:
            Ms[key] = sr[:, None] * C_Sk[key] if left else C_Sk[key] * sc[None, :]
        else:
            Ms[key] = C_Sk[key] * sc[None, :] if left else sr[:, None] * C_Sk[key]
    unitary_error_square = float(np.sum(S_ket**2) - sum(np.vdot(m, m).real for m in C_Sk.values()))     # :137
    if unitary_error_square < 0:
        err_mssg = (f"{mode.capitalize()} deviation from unitary: The square of the unitary error "
                    f"{unitary_error_square} is negative and exceeds the numerical tolerance {numerical_tol:.1e}.")
        assert_array_less(abs(unitary_error_square), numerical_tol, err_mssg)
        unitary_error = 0.0
    else:
        unitary_error = float(np.sqrt(unitary_error_square))
    logger.info(f"{mode.capitalize()} deviation from unitary: {unitary_error:.4e}")
    if unitary_error > unitary_tol:
        warnings.warn(f"\n{mode.capitalize()} overlap matrix deviates from unitarity by {unitary_error}.\n"
                      "Increasing the bond dimension may be useful.")
    # ---- U V of the SVD of every block: Jacobi with accumulated rotations on the zero-padded square block ----
    cplx = any(np.iscomplexobj(m) for m in Ms.values())
    dev = _Dev(device, cplx)
    keys = list(Ms)
    ps = [max(Ms[k].shape) for k in keys]
    off = np.concatenate(([0], np.cumsum([(p * p + 1) & ~1 for p in ps]))).astype(np.int64)
    host = np.zeros(off[-1] + 2, dev.np_dt)
    for k, p, o in zip(keys, ps, off[:-1]):
        pad = np.zeros((p, p), dev.np_dt)
        pad[: Ms[k].shape[0], : Ms[k].shape[1]] = Ms[k]
        host[o: o + p * p] = pad.reshape(-1, order="F")
    d_M, d_X = dev.up(host), dev.up(host.copy())
    d_W, d_V, d_G, d_U, d_VH, d_R = (dev.zeros(off[-1] + 2) for _ in range(6))
    d_s = dev.torch.zeros(int(sum(ps)) + 1, dtype=dev.torch.float64, device=dev.device)
    d_c = dev.torch.zeros(len(keys) + 1, dtype=dev.torch.int32, device=dev.device)
    el = dev.el
    jd = np.zeros(len(keys), nat.jacobi_desc)
    so = np.concatenate(([0], np.cumsum(ps)))
    gg, cn, cp, gr = [], np.zeros(len(keys), nat.colnorm_desc), [], []
    for i, (p, o) in enumerate(zip(ps, off[:-1])):
        scale = max(float(np.abs(Ms[keys[i]]).max()), 1e-300)
        jd[i] = (d_X.data_ptr() + el * o, d_W.data_ptr() + el * o, d_V.data_ptr() + el * o, d_s.data_ptr() + 8 * so[i],
                 d_c.data_ptr() + 4 * i, (1e-100 * scale) ** 2, p, p, p, p)
        gg.append((d_M.data_ptr() + el * o, d_V.data_ptr() + el * o, d_G.data_ptr() + el * o, p, p, p, p, p, p))
        cn[i] = (d_G.data_ptr() + el * o, d_U.data_ptr() + el * o, p, p, p, p, 0, 0)
        cp.append((d_V.data_ptr() + el * o, d_VH.data_ptr() + el * o, p, p, p, p, 3 if cplx else 1, 0))
        gr.append((d_U.data_ptr() + el * o, d_VH.data_ptr() + el * o, d_R.data_ptr() + el * o, p, p, p, p, p, p))
    if keys:
        t_j = dev.up(jd.view(np.uint8).reshape(-1))
        d_sw = dev.torch.zeros(len(keys), dtype=dev.torch.int32, device=dev.device)
        nat.check(dev.lib.tmf_jacobi_compact_batched(dev.dt, t_j.data_ptr(), len(keys), int(max(ps)), d_sw.data_ptr(),
                                                     dev.stream), "tmf_jacobi_compact_batched")
        dev.gemm(gg)                                               # G = M V = U S
        t_n = dev.up(cn.view(np.uint8).reshape(-1))
        nat.check(dev.lib.tmf_normalise_columns_batched(dev.dt, t_n.data_ptr(), len(keys), dev.stream), "normalise")
        dev.copy(cp)                                               # V^H
        dev.gemm(gr)                                               # rotation = U V^H
        dev.torch.cuda.synchronize(dev.device)
        nat.check_jacobi_sweeps(d_sw.cpu().numpy(), "Jacobi SVD of the overlap blocks (npc.svd, iMPS.py:158)")
    h_R = d_R.cpu().numpy()
    rot = {}
    for k, p, o in zip(keys, ps, off[:-1]):
        rot[k] = h_R[o: o + p * p].reshape(p, p).T[: Ms[k].shape[0], : Ms[k].shape[1]].copy()
    # ---- Schmidt value mixing (iMPS.py:173-183) ----
    err2 = 0.0
    for key, r in rot.items():
        sr = s_rows[rt[key[0]][0]: rt[key[0]][0] + rt[key[0]][1]]
        sc = s_cols[ct[key[1]][0]: ct[key[1]][0] + ct[key[1]][1]]
        if bra_side:
            Sb_C = sr[:, None] * r if left else r * sc[None, :]
        else:
            Sb_C = r * sc[None, :] if left else sr[:, None] * r
        err2 += float(np.linalg.norm(Sb_C - C_Sk[key]) ** 2)
    schmidt_error = float(np.sqrt(err2))
    logger.info(f"{mode.capitalize()} Schmidt value mixing:   {schmidt_error:.4e}")
    if schmidt_error > schmidt_tol:
        warnings.warn(f"\nMixing between unequal Schmidt value sectors on the {mode} side is\n"
                      f"{schmidt_error}. Increasing the number of sites may help.")
    return BlockMatrix(rot, overlap.rows, overlap.cols), unitary_error, schmidt_error


def MPS_to_iMPS(mps_short, mps_long, sites_per_cell: int, cut: int, unitary_tol: float = _UNITARY_TOL,
                schmidt_tol: float = _SCHMIDT_TOL, offset: Iterable[int | Literal["auto"]] | int | Literal["auto"] = "auto",
                unit_cell_width: int | None = None, *, device: str = "cuda:0", right: str = "rotate"):
    """Constructs an iMPS by comparing two finite MPS that differ by one unit cell (iMPS.py:232-441).
    Returns (iMPSData, iMPSError).

    right   "rotate" (iMPS.py:403-421): the last tensor is multiplied by the Procrustes rotation D of the right
            Schmidt-vector overlaps and the right-hand errors are reported.
            "project" (what ``C_to_iMPS`` does in the reference, slater.py:1508-1518, 1563 / pfaffian.py:2040-2091): the
            last tensor is expressed in the right Schmidt vectors of the SHORT chain, i.e. multiplied by the overlap
            matrix itself, and the right-hand errors are reported as zero."""
    if right not in ("rotate", "project"):
        raise ValueError(f"`right` must be 'rotate' or 'project', got {right!r}")
    from .gutzwiller import native

    came_as_tenpy = getattr(mps_short, "_temfpy_amd", None) is not None
    mps_short, mps_long = native(mps_short), native(mps_long)      # TeNPy objects this package returned
    L_short, L_long = mps_short.L, mps_long.L
    if L_short + sites_per_cell != L_long:
        raise ValueError("The given two MPS must differ by one unit cell, got "
                         f"{L_long} - {L_short} != {sites_per_cell}")
    def charge_kind(m):      # what the ChargeInfo of the TeNPy object would be
        if hasattr(m, "bonds"):
            return "N" if hasattr(m.bonds[0], "q_left") else "parity"
        return "spin " + str(getattr(m, "conserve", None))               # gutzwiller.SpinMPSData
    if charge_kind(mps_short) != charge_kind(mps_long):
        raise ValueError("Incompatible ChargeInfo in the two MPS")          # iMPS.py:312-313
    assert all(x is not None for x in mps_short.form), "mps_short is not canonical"
    assert all(x is not None for x in mps_long.form), "mps_long is not canonical"
    if unit_cell_width is None:
        cyl1, cyl2 = (m.L // m.unit_cell_width for m in (mps_short, mps_long))   # N_sites_per_hor_spacing
        if cyl1 != cyl2:
            warnings.warn(f"Unequal cylinder circumferences {cyl1}, {cyl2},\ndiscard `unit_cell_width` of input MPS")
            cyl1 = cyl2 = 1
        if cut % cyl1 != 0:
            warnings.warn(f"{cut = } not divisible into cylinder circumferences of {cyl1},\n"
                          "discard `unit_cell_width` of input MPS")
            cyl1 = cyl2 = 1
        unit_cell_width = sites_per_cell // cyl1
    else:
        assert sites_per_cell % unit_cell_width == 0, f"{unit_cell_width = } does not divide {sites_per_cell = }"
        cyl1 = sites_per_cell // unit_cell_width
        assert cut % cyl1 == 0, f"{cut = } not divisible into requested cylinder circumferences of {cyl1}"
    cplx = any(np.iscomplexobj(m.sites[0].dense() if hasattr(m, "sites") else m.dense_tensors()[0]) for m in (mps_short, mps_long))
    dev = _Dev(device, cplx)
    cs, cl_ = _Chain(dev, mps_short), _Chain(dev, mps_long)
    mod = cs.mod
    if cl_.mod != mod:
        raise ValueError("Incompatible ChargeInfo in the two MPS")
    S0, q0 = cs.lam[cut], np.asarray(cs.f.charges[cut], np.int64)
    if isinstance(offset, Iterable) and not isinstance(offset, str):
        offset = list(offset)
        assert len(offset) == 1, "Expected 1 offsets"
        offset = offset[0]
    if isinstance(offset, (int, np.integer)):
        offset = int(offset)
    elif offset == "auto":                                                     # iMPS.py:360-366
        offset = 0 if mod else int(round(float((S0**2) @ q0)))
    else:
        raise TypeError(f"Expected integer or 'auto' as offset, got {offset!r}")
    logger.info("Using charge offsets %s", offset)
    # left gauge fixing matrix C, right gauge fixing matrix D (iMPS.py:381-413)
    C0 = _overlap(dev, cs, 0, cut, cl_, 0, cut, "left")
    C, left_unitary, left_schmidt = basis_rotation(C0, S0, cl_.lam[cut], mode="left", unitary_tol=unitary_tol,
                                                   schmidt_tol=schmidt_tol, device=device)
    D0 = _overlap(dev, cs, cut, L_short, cl_, cut + sites_per_cell, L_long, "right")
    if right == "rotate":
        D, right_unitary, right_schmidt = basis_rotation(D0, S0, cl_.lam[cut + sites_per_cell], mode="right",
                                                         unitary_tol=unitary_tol, schmidt_tol=schmidt_tol, device=device)
    else:
        D, right_unitary, right_schmidt = D0, 0.0, 0.0
    # unit cell in right canonical form, gauge unitaries on the first and last tensor (iMPS.py:415-421)
    cell = [cl_.to_form(cut + i, "B") for i in range(sites_per_cell)]
    dq = int(cl_.f.charges[L_long][0]) - int(cs.f.charges[L_short][0])
    dq = dq % mod if mod else dq
    shift = (lambda c: (c - dq) % mod) if mod else (lambda c: c - dq)
    upC = {k: dev.up(np.asfortranarray(v).astype(dev.np_dt).reshape(-1, order="F")) for k, v in C.blocks.items()}
    upD = {k: dev.up(np.asfortranarray(v).astype(dev.np_dt).reshape(-1, order="F")) for k, v in D.blocks.items()}
    g, first = [], {}
    for (p, cl), (ptr, n_l, n_r, cr) in cell[0].items():
        key = (cl, cl)                                             # C: rows short (charge cl), columns long (cl)
        if key not in upC:
            continue
        ns = C.blocks[key].shape[0]
        t = dev.zeros(ns * n_r)
        g.append((upC[key].data_ptr(), ptr, t.data_ptr(), ns, n_r, n_l, ns, n_l, ns))
        first[(p, cl)] = [t.data_ptr(), ns, n_r, cr]
    dev.gemm(g)
    cell[0] = first
    g, last = [], {}
    for (p, cl), (ptr, n_l, n_r, cr) in cell[-1].items():
        key = (cr, shift(cr))                                      # D: rows long (charge cr), columns short (cr - dq)
        if key not in upD:
            continue
        ns = D.blocks[key].shape[1]
        t = dev.zeros(n_l * ns)
        g.append((ptr, upD[key].data_ptr(), t.data_ptr(), n_l, ns, n_r, n_l, n_r, n_l))
        last[(p, cl)] = [t.data_ptr(), n_l, ns, shift(cr)]
    dev.gemm(g)
    cell[-1] = last
    dev.torch.cuda.synchronize(dev.device)
    # ---- results -----------------------------------------------------------------------------------------------
    lam = [S0] + [cl_.lam[cut + i] for i in range(1, sites_per_cell)] + [S0]
    q_b = [q0] + [np.asarray(cl_.f.charges[cut + i], np.int64) for i in range(1, sites_per_cell)] + [q0]
    tabs = [_sector_table(q) for q in q_b]
    blocks = []
    for i, S in enumerate(cell):
        bl = []
        for (p, cl), (ptr, n_l, n_r, cr) in S.items():
            if cl not in tabs[i] or cr not in tabs[i + 1]:
                continue
            buf = _read(dev, ptr, n_l * n_r).reshape(n_r, n_l).T
            (l0, nl), (r0, nr) = tabs[i][cl], tabs[i + 1][cr]
            assert (nl, nr) == (n_l, n_r)
            lab = (lambda c: (c - offset) % mod) if mod else (lambda c: c - offset)
            bl.append((p, lab(cl), lab(cr), l0, l0 + nl, r0, r0 + nr, buf))
        blocks.append(bl)
    res = iMPSData(blocks, lam, [(q - offset) % mod if mod else q - offset for q in q_b], dq, unit_cell_width,
                   conserve=charge_kind(mps_short))
    if came_as_tenpy:          # TeNPy in, TeNPy out (iMPS.py:430-441)
        try:
            res = res.to_tenpy()
        except ImportError:
            pass
    return res, iMPSError(left_unitary, left_schmidt, right_unitary, right_schmidt)


def cell_from_determinants(shard, L_long: int, cut: int, sites_per_cell: int, offset: int, unit_cell_width: int, *,
                           unitary_tol: float = _UNITARY_TOL, schmidt_tol: float = _SCHMIDT_TOL, device: str = "cuda:0"):
    """Unit cell of ``slater.C_to_iMPS`` the reference's way (slater.py:1499-1563) from ONE sweep over the cuts
    ``cut .. cut + sites_per_cell - 1`` of the long chain and the cut of the short chain (``Engine.run_gen(..., second=...)``):
    right-canonical tensors of the long chain, the last one with the right Schmidt vectors of the SHORT chain as bra
    (:1513-1514), and on the first one the Procrustes rotation of the overlaps of the left Schmidt vectors, which are
    determinants between two bases of the same orbitals - no physical leg (:1538-1553, :1023-1024).  No environments, no
    full conversion of either chain.  Returns (iMPSData, iMPSError) with zero right-hand errors (:1563)."""
    KEY = L_long + 1
    n_reg = sites_per_cell - 1
    b_short = shard.bond(KEY + cut + sites_per_cell)                     # cut of the short chain (embedded, see C_to_iMPS)
    b_long = [shard.bond(cut + i) for i in range(sites_per_cell)]
    bonds = [b_short] + b_long[1:] + [b_short]                            # slater.py:1502, :1515, :1520
    lam = [np.array(b.lam) for b in bonds]
    q_b = [np.asarray(b.q_left, np.int64) for b in bonds]
    n_of = lambda b: int(b.n_filled_left + b.n_filled_right + len(b.e))   # noqa: E731
    dq = n_of(b_long[0]) - n_of(b_short)                                  # particles per unit cell (qtotal of the last tensor, :1092)
    site0 = int(shard.meta["s_lo"])
    sds = [shard.site(site0 + j) for j in range(n_reg + 2)]              # regular sites, last tensor, gauge overlaps
    gauge = sds[-1]

    def right_blocks(sd, q_l, q_r):
        """(p, q_l, q_r, l0, l1, r0, r1, (n_l x n_r) array) of a right-mode site: rows of its blocks are the merged (p, vR)
        leg sorted by charge, columns the vL states."""
        out = []
        for (_, r0, r1, c0, c1, arr) in sd.blocks:
            ps, al = np.asarray(sd.bra_p[r0:r1]), np.asarray(sd.bra_alpha[r0:r1])
            for p in (0, 1):
                sel = np.nonzero(ps == p)[0]
                if sel.size == 0:
                    continue
                a0, a1 = int(al[sel[0]]), int(al[sel[-1]]) + 1
                assert np.array_equal(al[sel], np.arange(a0, a1)), "vR states of a charge block are not contiguous"
                out.append((p, int(q_l[c0]), int(q_r[a0]), c0, c1, a0, a1, np.ascontiguousarray(arr[sel].T)))
        return out

    q_long = [np.asarray(b.q_left, np.int64) for b in b_long]
    cell = [right_blocks(sds[i], q_long[i], q_long[i + 1] if i + 1 < sites_per_cell else q_b[-1]) for i in range(sites_per_cell)]
    # gauge matrix C[a, b] = <L^short_a | L^long_b> (rows short, columns long; equal charges)
    gblocks = {}
    for (_, r0, r1, c0, c1, arr) in gauge.blocks:
        al = np.asarray(gauge.bra_alpha[r0:r1])
        assert np.array_equal(al, np.arange(al[0], al[0] + len(al)))
        gblocks[(int(q_b[0][al[0]]), int(q_long[0][c0]))] = np.array(arr)
    C0 = BlockMatrix(gblocks, q_b[0], q_long[0])
    # (the reference passes the unnormalised Schmidt values of the two cuts, slater.py:1542-1543)
    C, left_unitary, left_schmidt = basis_rotation(C0, b_short.lam_raw, b_long[0].lam_raw, mode="left", unitary_tol=unitary_tol,
                                                   schmidt_tol=schmidt_tol, device=device)
    cplx = any(np.iscomplexobj(b[7]) for bl in cell for b in bl) or any(np.iscomplexobj(v) for v in C.blocks.values())
    dev = _Dev(device, cplx)
    rt = _sector_table(q_b[0])
    g, first = [], []
    for (p, ql, qr, l0, l1, r0, r1, arr) in cell[0]:                      # slater.py:1552: C . B_0 on the vL leg
        key = (ql, ql)
        if key not in C.blocks:
            continue
        cb = C.blocks[key]
        ns, nl, nr = cb.shape[0], cb.shape[1], arr.shape[1]
        assert nl == arr.shape[0]
        t = dev.zeros(ns * nr)
        a_ = dev.up(np.asfortranarray(cb).astype(dev.np_dt).reshape(-1, order="F"))
        b_ = dev.up(np.asfortranarray(arr).astype(dev.np_dt).reshape(-1, order="F"))
        g.append((a_.data_ptr(), b_.data_ptr(), t.data_ptr(), ns, nr, nl, ns, nl, ns))
        first.append((p, ql, qr, rt[ql][0], rt[ql][0] + ns, r0, r1, t, ns, nr))
    dev.gemm(g)
    dev.torch.cuda.synchronize(dev.device)
    cell[0] = [(p, ql, qr, l0, l1, r0, r1, t.cpu().numpy()[: ns * nr].reshape(nr, ns).T.copy())
               for (p, ql, qr, l0, l1, r0, r1, t, ns, nr) in first]
    blocks = [[(p, ql - offset, qr - offset, l0, l1, r0, r1, a) for (p, ql, qr, l0, l1, r0, r1, a) in bl] for bl in cell]
    res = iMPSData(blocks, lam, [q - offset for q in q_b], dq, unit_cell_width, conserve="N")
    res.gauge_overlaps = C0          # <L^short_a | L^long_b> before the rotation (diagnostics, tests)
    return res, iMPSError(left_unitary, left_schmidt, 0.0, 0.0)


def _read(dev, ptr, count):
    """Host copy of `count` elements at device address `ptr` (inside one of the tensors `dev` keeps alive)."""
    for t in dev.keep:
        lo = t.data_ptr()
        hi = lo + t.numel() * t.element_size()
        if lo <= ptr < hi and t.dtype in (dev.torch.float64, dev.torch.complex128):
            o = (ptr - lo) // dev.el
            return t[o: o + count].cpu().numpy()
    raise KeyError(ptr)


__all__ = ["MPS_to_iMPS", "overlap_schmidt", "basis_rotation", "iMPSError", "iMPSData", "BlockMatrix", "cell_from_determinants"]
