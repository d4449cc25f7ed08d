"""Host driver of the MI355X Slater -> MPS sweep.

The reference walks the chain site by site (slater.py:1300-1346) and calls LAPACK per cut.
All L+1 cuts and all L sites are independent given C, so here every stage is ONE batched
launch over all of them (descriptor arrays, variable sizes), with only two host
round-trips: the entangled eigenvalues come back for the best-first enumeration
(integer work, C++ on the host, overlapped with stage F), and the index lists go up for the
determinant stage.

Per cut side (block A = C_LL or C_RR, off-diagonal block F = C_LR or C_RL), using that C is a
projector (A - A^2 = F F^H, so entangled orbitals = left singular vectors of F with
sigma^2 = e (1 - e) >= cutoff (1 - cutoff), the same set as slater.py:350):
  E1  Y = F Omega                      running sums over the nested blocks of all cuts (nested.hip);
                                       randomised range finder, 64 columns (128 / 256 when a cut needs it)
  E2  Q = qr(Y)                        blocked Gram-Schmidt driven from C++ (bcgs.hip): MFMA projections
                                       + LDS panel kernel that drops columns of rounding noise
  E3  B^H = F^H Q ; R^H = (B^H)^H qr(B^H).Q     GEMM + Gram-Schmidt + GEMM
  E4  R^H = U diag(sigma) V^H          one-sided Jacobi in LDS, left vectors only (relative accuracy
                                       near sigma = 1e-6); the smallest captured sigma is checked
  E5  U0 = Q U[:, sigma^2 >= thr]      GEMM
  E6  T = U0^H A U0 ; T X = X diag(e)  2 GEMMs + Jacobi   (Rayleigh-Ritz: eigenpairs of A)
  E7  U_E = U0 X                       GEMM
  F   filled basis: orthonormalise (1 - U_E U_E^H) A Omega_f   (nested running sums + Gram-Schmidt
                                       with Cholesky-QR panels); centre right orbitals = C_RL v_L
Per site:
  S1  O = V_bra^H V_ket                MFMA GEMM  (slater.py:1071)
  S2  W = signed gather of O           [always block | sometimes], physical orbital row
  S3  det_always, Schur complement     blocked LU in LDS panels (slater.py:1077-1090)
  S4  all minors of all sectors        one pivoted exchange of the sector matrix per workgroup, every
                                       minor a determinant of order d <= 4 of it (det_ppt.hip;
                                       slater.py:828-869)
Self-check (testing.py:131-177): reconstruction deviations of the centre cut (recon.hip).
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np

from . import _native as nat
from .mps_data import MPSData, ShardArrays

logger = logging.getLogger("temfpy_amd.slater")

P_RANGE = 64     # columns of the range finder = LDS limit of the Jacobi kernel (complex128)
PANEL_W = 16


def _cdiv(a, b):
    return (a + b - 1) // b


class _GpuWait:
    """Wait handle for everything enqueued so far on the current stream."""

    def __init__(self, torch, device):
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(device))

    def ready(self):
        return self.ev.query()

    def block(self):
        self.ev.synchronize()


class _EventWait:
    def __init__(self, ev):
        self.ev = ev

    def ready(self):
        return self.ev.query()

    def block(self):
        self.ev.synchronize()


class PinnedSink:
    """Host memory of a result: page-locked blocks from torch's caching host allocator (the block of a
    released result is reused by the next conversion; a pageable destination ran at ~6 GB/s)."""

    def __init__(self, torch):
        self.torch = torch

    def alloc(self, nbytes):
        t = self.torch.empty(int(nbytes), dtype=self.torch.uint8, pin_memory=True)
        return t.numpy(), t


class _ThreadWait:
    def __init__(self, th):
        self.th = th

    def ready(self):
        return not self.th.is_alive()

    def block(self):
        self.th.join()


def _drive(gen):
    """Runs a sweep generator to completion, blocking at every wait point (the ordinary, unpipelined call)."""
    try:
        while True:
            next(gen).block()
    except StopIteration as stop:
        return stop.value


class Engine:
    def __init__(self, device="cuda:0", profile=None):
        import torch

        if not torch.cuda.is_available():
            raise nat.NativeError("temfpy_amd needs a HIP device (torch.cuda.is_available() is False); "
                                  "there is no CPU fallback")
        self.torch = torch
        self.device = torch.device(device)
        self.lib = nat.load()
        self.profile = bool(int(os.environ.get("TMF_PROFILE", "0"))) if profile is None else profile
        self.range_iterations_used, self.range_width, self.range_floor = 0, P_RANGE, 0.0
        self.checks = True          # evaluate testing.check_schmidt_decomposition's deviations on the device
        self.check_results = {}
        self.timings = {}
        self._keep = []  # descriptor tensors must outlive the launches that read them
        self._pin_t = self._pin_np = self._dev_arena = None
        self._pin_cap = self._pin_off = 0
        self.force_direct_det = bool(int(os.environ.get("TMF_DIRECT_DET", "0")))  # A/B switch
        self.time_gemm = False   # bench.py: HIP events around every MFMA GEMM launch
        self.gemm_events = []
        self._sweep_max = []     # device scalars: largest Jacobi sweep count of every launch since the last check
        self._inflight = []      # (event, references) of downloads still running on the copy stream
        self._copy_stream = torch.cuda.Stream(device=self.device)   # device -> host: the tensors of a result
        self._up_stream = torch.cuda.Stream(device=self.device)     # host -> device: index lists
        self._sink = None
        self._ctx = None         # tmf_ctx of the C++ sweep (created on first use)
        self._tickets = []       # (download ticket, host buffer) of results still being copied
        self.kernel_info = None
        self.coord = None        # multi-rank runs: object with .max(np.ndarray) -> elementwise max over the ranks

    # ------------------------------------------------------------------ plumbing
    @property
    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _up(self, a: np.ndarray):
        """Descriptor upload: host memcpy into a pinned staging arena, asynchronous copy into the
        mirrored device arena on the launch stream (a pageable `.to(device)` blocks the host and
        drains the stream: 235 us x 344 uploads per conversion, measured)."""
        a = np.ascontiguousarray(a)
        nb = a.nbytes
        off = (self._pin_off + 255) & ~255
        if self._pin_t is None or off + nb > self._pin_cap:
            cap = max(2 * self._pin_cap, 2 * nb, 64 << 20)
            if self._pin_t is not None:
                self._keep.append((self._pin_t, self._dev_arena))  # still referenced by queued copies
            self._pin_t = self.torch.empty(cap, dtype=self.torch.uint8, pin_memory=True)
            self._pin_np = self._pin_t.numpy()
            self._dev_arena = self.torch.empty(cap, dtype=self.torch.uint8, device=self.device)
            self._pin_cap, off = cap, 0
        self._pin_np[off: off + nb] = a.view(np.uint8).reshape(-1)
        dst = self._dev_arena[off: off + nb]
        dst.copy_(self._pin_t[off: off + nb], non_blocking=True)
        self._pin_off = off + nb
        return dst

    def _alloc(self, count, real=False, zero=False):
        dt = self.torch.float64 if (real or self.dtype == nat.TMF_F64) else self.torch.complex128
        f = self.torch.zeros if zero else self.torch.empty
        return f(max(int(count), 1), dtype=dt, device=self.device)

    def _tick(self, name, t0):
        if self.profile:
            self.torch.cuda.synchronize(self.device)
        self.timings[name] = self.timings.get(name, 0.0) + time.perf_counter() - t0

    # ------------------------------------------------------------------ batched ops
    def gemm(self, opA, alpha, beta, A, B, C, M, N, K, lda, ldb, ldc):
        """Batched C = alpha op(A) B + beta C over arrays of problems (device addresses)."""
        M, N, K = (np.asarray(x, np.int64) for x in (M, N, K))
        sel = np.nonzero((M > 0) & (N > 0))[0]
        if sel.size == 0:
            return
        d = np.zeros(sel.size, nat.gemm_desc)
        for f, v in (("A", A), ("B", B), ("C", C), ("M", M), ("N", N), ("K", K), ("lda", lda), ("ldb", ldb), ("ldc", ldc)):
            d[f] = np.broadcast_to(np.asarray(v), M.shape)[sel]
        d["lda"] = np.maximum(d["lda"], 1)
        d["ldb"] = np.maximum(d["ldb"], 1)
        tn = 16 if int(d["N"].max()) <= 16 else 64
        tm = _cdiv(d["M"].astype(np.int64), 64)
        tnn = _cdiv(d["N"].astype(np.int64), tn)
        cnt = tm * tnn
        total = int(cnt.sum())
        prob = np.repeat(np.arange(sel.size), cnt)
        local = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        tiles = np.zeros((total, 4), np.int32)
        tiles[:, 0] = prob
        tiles[:, 1] = local % tm[prob]
        tiles[:, 2] = local // tm[prob]
        order = np.argsort(-d["K"][prob].astype(np.int64), kind="stable")  # longest tiles first
        tiles = tiles[order]
        dd, dt = self._up(d), self._up(tiles)
        if self.time_gemm:
            ev0, ev1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            ev0.record(self.torch.cuda.current_stream(self.device))
        nat.check(self.lib.tmf_gemm_batched(self.dtype, opA, float(alpha), float(beta), dd.data_ptr(), dt.data_ptr(),
                                            total, tn, self.stream), "tmf_gemm_batched")
        if self.time_gemm:
            ev1.record(self.torch.cuda.current_stream(self.device))
            fl = float((d["M"].astype(np.float64) * d["N"] * d["K"]).sum()) * (8.0 if self.dtype == nat.TMF_C128 else 2.0)
            self.gemm_events.append((ev0, ev1, fl))

    def bcgs2(self, base, rows, ld, c_begin, c_end, scratch_ptr, passes=2, cholqr=False):
        """Orthonormalise columns [c_begin, c_end) of every matrix against all columns before
        them (blocked classical Gram-Schmidt with re-orthogonalisation, tmf_bcgs_batched: the panel
        loop and its descriptors live in C++ / on the device).
        passes=2 for well-conditioned slabs; passes=3 for numerically rank-deficient ones
        (range finder), where a second pass still acts on rounding noise.  cholqr: Cholesky-QR inside the
        panels (well-conditioned slabs only: the filled-orbital bases)."""
        base, rows, ld, c_begin, c_end, scratch_ptr = (np.asarray(x, np.int64) for x in
                                                       (base, rows, ld, c_begin, c_end, scratch_ptr))
        keep = np.nonzero((rows > 0) & (c_end > c_begin))[0]
        if keep.size == 0:
            return
        keep = keep[np.argsort(-rows[keep], kind="stable")]      # long slabs first: their tiles run longest
        base, rows, ld, c_begin, c_end, scratch_ptr = (x[keep] for x in (base, rows, ld, c_begin, c_end, scratch_ptr))
        span = c_end - c_begin
        # norms of the raw columns: the panel kernel zeroes columns whose residual is rounding noise
        noff = np.concatenate(([0], np.cumsum(span)))[:-1]
        d_nrm = self.torch.zeros(max(int(span.sum()), 1), dtype=self.torch.float64, device=self.device)
        self._keep.append(d_nrm)
        nrmp = d_nrm.data_ptr() + 8 * noff
        nd = np.zeros(base.size, nat.norms_desc)
        nd["src"], nd["out"] = base + c_begin * ld * self.elem, nrmp
        nd["n"], nd["c"], nd["lds_"] = rows, span, ld
        t_nd = self._up(nd)
        nat.check(self.lib.tmf_column_norms_batched(self.dtype, t_nd.data_ptr(), base.size, self.stream), "norms")
        bd = np.zeros(base.size, nat.bcgs_desc)
        bd["base"], bd["scratch"], bd["norms"] = base, scratch_ptr, nrmp
        bd["rows"], bd["ld"], bd["c_begin"], bd["c_end"] = rows, ld, c_begin, c_end
        t_bd = self._up(bd)
        wb = int(self.lib.tmf_bcgs_work_bytes(nat._p(bd), base.size))
        d_work = self.torch.empty(wb, dtype=self.torch.uint8, device=self.device)
        self._keep.append(d_work)
        nat.check(self.lib.tmf_bcgs_batched(self.dtype, t_bd.data_ptr(), nat._p(bd), base.size, passes,
                                            1 if cholqr else 0, d_work.data_ptr(), wb, self.stream), "tmf_bcgs_batched")

    def house_slab(self, base, rows, ld, cols, inplace=True, r_only=None, r_ld=None):
        """Householder QR of tall slabs (tmf_house_slab_batched: columns in registers, reflector blocks in LDS, one
        workgroup per slab): the thin orthonormal factor of every matrix base[i] (rows[i] x cols[i], leading
        dimension ld[i]).  Orthogonal for any numerical rank - no projection passes, no per-column rank decision.
        inplace: Q overwrites the slab.  Otherwise Q stays in the scratch the kernel builds it in (saves one pass
        over every slab) and the array of its addresses is returned (leading dimension = rows); slabs without
        rows or columns keep their address."""
        base, rows, ld, cols = (np.asarray(x, np.int64) for x in (base, rows, ld, cols))
        out = base.copy()
        keep = np.nonzero((rows > 0) & (cols > 0))[0]
        if keep.size == 0:
            return out
        keep = keep[np.argsort(-rows[keep], kind="stable")]      # long slabs first
        b_, r_, l_, c_ = (x[keep] for x in (base, rows, ld, cols))
        d = np.zeros(b_.size, nat.slab_desc)
        if r_only is not None:      # only the conjugate transpose of the triangular factor is wanted (slab destroyed)
            d["A"], d["Q"], d["R"] = b_, 0, np.asarray(r_only, np.int64)[keep]
            d["n"], d["c"], d["lda"], d["ldq"], d["flags"] = r_, c_, l_, 1, 1 | 4
            d["ldr"] = np.asarray(r_ld, np.int64)[keep]
            t_d = self._up(d)
            nat.check(self.lib.tmf_house_slab_batched(self.dtype, t_d.data_ptr(), b_.size, int(r_.max()), int(c_.max()),
                                                      self.stream), "tmf_house_slab_batched")
            return out
        sizes = r_ * c_
        off = np.concatenate(([0], np.cumsum((sizes + 1) & ~1)))
        d_q = self._alloc(int(off[-1]) + 2)
        self._keep.append(d_q)
        d["A"], d["Q"], d["R"] = b_, d_q.data_ptr() + off[:-1] * self.elem, 0
        d["n"], d["c"], d["lda"], d["ldq"], d["ldr"], d["flags"] = r_, c_, l_, r_, 1, 0 if inplace else 2
        t_d = self._up(d)
        nat.check(self.lib.tmf_house_slab_batched(self.dtype, t_d.data_ptr(), b_.size, int(r_.max()), int(c_.max()),
                                                  self.stream), "tmf_house_slab_batched")
        if not inplace:
            out[keep] = d["Q"].astype(np.int64)
        return out

    def house_general(self, base, rows, ld, cols):
        """Householder QR in place of slabs of any width (tmf_house_qr_batched: matrix in global memory, one workgroup per
        slab): the range finders wider than the slab kernel's 64 columns.  Orthogonal for any rank, where the blocked
        Gram-Schmidt lost an entangled orbital of a spinful chain with exactly decoupled species (tests/soak/soak_small.py seed 30023)."""
        base, rows, ld, cols = (np.asarray(x, np.int64) for x in (base, rows, ld, cols))
        keep = np.nonzero((rows > 0) & (cols > 0))[0]
        if keep.size == 0:
            return
        mm, mn = int(rows[keep].max()), int(cols[keep].max())
        if (os.environ.get("TMF_WIDE_QR") != "global" and mm <= (1024 if self.elem == 16 else 2048)
                and (mm * 8 + mn + 4) * self.elem + 64 <= 150 * 1024):
            # rows that fit the panel kernel: factored there, the Q's formed in place afterwards (csrc/sweep.cpp house_general)
            tau_sz = (cols[keep] + 1) & ~1
            d_tau = self._alloc(int(tau_sz.sum()) + 2)
            sd = np.zeros(keep.size, nat.slab_desc)
            sd["A"], sd["R"], sd["n"], sd["c"], sd["lda"], sd["ldq"], sd["ldr"], sd["flags"] = base[keep], 0, rows[keep], cols[keep], ld[keep], 1, 1, 8
            sd["Q"] = d_tau.data_ptr() + self.elem * (np.cumsum(tau_sz) - tau_sz)
            sd = sd[np.argsort(-(sd["n"].astype(np.int64) * sd["c"]), kind="stable")]
            t_d = self._up(sd)
            nat.check(self.lib.tmf_house_slab_batched(self.dtype, t_d.data_ptr(), keep.size, mm, mn, self.stream), "tmf_house_slab_batched")
            nat.check(self.lib.tmf_house_form_q_batched(self.dtype, t_d.data_ptr(), keep.size, mm, mn, self.stream), "tmf_house_form_q_batched")
            return
        keep = keep[np.argsort(-(rows[keep] * cols[keep]), kind="stable")]
        d = np.zeros(keep.size, nat.qr_desc)
        d["A"], d["R"], d["m"], d["n"], d["lda"], d["ldr"], d["flags"] = base[keep], 0, rows[keep], cols[keep], ld[keep], 1, 0
        t_d = self._up(d)
        nat.check(self.lib.tmf_house_qr_batched(self.dtype, t_d.data_ptr(), keep.size, int(rows[keep].max()), int(cols[keep].max()),
                                                self.stream), "tmf_house_qr_batched")

    def jacobi(self, X, V, s, count, thresh2, p, ldx, ldv, left_only=False):
        """One-sided Jacobi per problem.  ``left_only``: ``V`` receives the normalised LEFT singular
        vectors (tmf_svd_left_batched, no rotation accumulator) instead of the right ones.
        p > 64: block variant in global memory (tmf_jacobi_block_batched; destroys X)."""
        p = np.asarray(p, np.int64)
        sel = np.nonzero(p > 0)[0]
        if sel.size == 0:
            return
        d = np.zeros(sel.size, nat.jacobi_desc)
        big = int(p.max()) > 64
        out_field = "U" if (left_only or big) else "V"
        for f, v in (("X", X), (out_field, V), ("s", s), ("count", count), ("p", p), ("ldx", ldx),
                     ("ldu" if out_field == "U" else "ldv", ldv)):
            d[f] = np.broadcast_to(np.asarray(v), p.shape)[sel]
        d["thresh2"] = thresh2
        if out_field == "V":
            d["ldu"] = 1
        # sweep count of every problem: the cap (60) means "not converged" and is checked at the next host round trip
        d_sw = self.torch.zeros(sel.size, dtype=self.torch.int32, device=self.device)
        self._keep.append(d_sw)
        if big:
            if not left_only:   # workspace for the accumulated rotations
                pp = p[sel]
                wo = np.concatenate(([0], np.cumsum(pp * pp)))
                d_ws = self._alloc(int(wo[-1]))
                self._keep.append(d_ws)
                d["V"], d["ldv"] = d_ws.data_ptr() + wo[:-1] * self.elem, pp
            dd = self._up(d)
            nat.check(self.lib.tmf_jacobi_block_batched(self.dtype, 0 if left_only else 1, dd.data_ptr(), sel.size,
                                                        int(p.max()), d_sw.data_ptr(), self.stream), "tmf_jacobi_block_batched")
        else:
            dd = self._up(d)
            fn = self.lib.tmf_svd_left_batched if left_only else self.lib.tmf_jacobi_batched
            nat.check(fn(self.dtype, dd.data_ptr(), sel.size, int(p.max()), d_sw.data_ptr(), self.stream),
                      "tmf_svd_left_batched" if left_only else "tmf_jacobi_batched")
        self._sweep_max.append(d_sw.amax())
        if os.environ.get("TMF_JACOBI_SWEEPS"):  # debugging aid: sweep statistics of every launch
            h = d_sw.cpu().numpy()
            print(f"jacobi(left_only={left_only}) p<= {int(p.max())}: sweeps min {h.min()} mean {h.mean():.1f} "
                  f"max {h.max()}; hist {np.bincount(h).tolist()}", flush=True)

    def nested_products(self, kind, D, Cp, Omp, ldo, x, side, dest, ncol, ld):
        """Y_i = A_i Omega (kind "A") or F_i Omega (kind "F") for all cut sides at once through
        running sums over the shared index (tmf_nested_products_batched).  x: cut position in matrix
        indices, side 0: the block left of x, side 1: right of x; Omega rows carry global indices."""
        x, side, dest, ncol, ld = (np.asarray(a) for a in (x, side, dest, ncol, ld))
        descs = np.zeros(2, nat.nested_desc)
        nd = 0
        for sd_ in (0, 1):
            sel = np.nonzero((side == sd_) & (ncol > 0))[0]
            if sel.size == 0:
                continue
            tab_d, tab_n, tab_l = np.zeros(D + 1, np.uint64), np.zeros(D + 1, np.int32), np.ones(D + 1, np.int32)
            tab_d[x[sel]], tab_n[x[sel]], tab_l[x[sel]] = dest[sel], ncol[sel], ld[sel]
            t_d, t_n, t_l = self._up(tab_d), self._up(tab_n), self._up(tab_l)
            suffix = (sd_ == 1) if kind == "A" else (sd_ == 0)
            descs[nd] = (Cp, Omp, t_d.data_ptr(), t_n.data_ptr(), t_l.data_ptr(), D, D, ldo, int(suffix), sd_,
                         int(x[sel].min()), int(x[sel].max()), int(ncol[sel].max()))
            nd += 1
        if nd == 0:
            return
        t_desc = self._up(descs[:nd])
        nat.check(self.lib.tmf_nested_products_batched(self.dtype, t_desc.data_ptr(), nd, D,
                                                       int(descs["maxc"][:nd].max()), self.stream),
                  "tmf_nested_products_batched")

    def recon_errors(self, items):
        """Launches tmf_recon_error_batched for a list of checks; returns the device tensor that will
        hold one deviation per item.  item = dict(T, X, Y, w (host float64 array or None), rows, cols,
        q, inner, ldt, ldx, ldy, mode, y_reverse)."""
        nchk = len(items)
        d_dev = self.torch.zeros(max(nchk, 1), dtype=self.torch.float64, device=self.device)
        self._keep.append(d_dev)
        if nchk == 0:
            return d_dev
        ws = [np.asarray(it["w"], np.float64) for it in items if it.get("w") is not None]
        w_off, t_w = {}, None
        if ws:
            o = 0
            for i, it in enumerate(items):
                if it.get("w") is not None:
                    w_off[i] = o
                    o += len(it["w"])
            t_w = self._up(np.concatenate(ws + [np.zeros(1)]))
        d = np.zeros(nchk, nat.recon_desc)
        col = lambda key, default=0: np.array([it.get(key, default) for it in items])  # noqa: E731
        d["T"], d["X"], d["Y"] = col("T"), col("X"), col("Y")
        d["w"] = [0 if i not in w_off else t_w.data_ptr() + 8 * w_off[i] for i in range(nchk)]
        d["out"] = d_dev.data_ptr() + 8 * np.arange(nchk)
        for key, default in (("rows", 0), ("cols", 0), ("q", 0), ("inner", 0), ("ldt", 1), ("ldx", 1), ("ldy", 1),
                             ("mode", 0), ("y_reverse", 0)):
            d[key] = col(key, default)
        # tile table (problem, tile_row, tile_col) of all items at once
        tr, tc = _cdiv(d["rows"].astype(np.int64), 64), _cdiv(d["cols"].astype(np.int64), 64)
        cnt = tr * tc
        prob = np.repeat(np.arange(nchk), cnt)
        local = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        tiles = np.ascontiguousarray(np.stack((prob, local // np.maximum(tc[prob], 1), local % np.maximum(tc[prob], 1)), axis=1),
                                     np.int32)
        if len(tiles) == 0:
            return d_dev
        t_d, t_t = self._up(d), self._up(tiles)
        nat.check(self.lib.tmf_recon_error_batched(self.dtype, t_d.data_ptr(), t_t.data_ptr(), len(tiles), self.stream),
                  "tmf_recon_error_batched")
        return d_dev

    def colcopy(self, src, dst, n, c, lds_, ldd, reverse=0, flip_odd=0):
        n, c = np.asarray(n, np.int64), np.asarray(c, np.int64)
        sel = np.nonzero((n > 0) & (c > 0))[0]
        if sel.size == 0:
            return
        d = np.zeros(sel.size, nat.colnorm_desc)
        for f, v in (("src", src), ("dst", dst), ("n", n), ("c", c), ("lds_", lds_), ("ldd", ldd)):
            d[f] = np.broadcast_to(np.asarray(v), n.shape)[sel]
        d["reverse"], d["flip_odd"] = reverse, flip_odd
        dd = self._up(d)
        nat.check(self.lib.tmf_normalise_columns_batched(self.dtype, dd.data_ptr(), sel.size, self.stream),
                  "tmf_normalise_columns_batched")

    def entangled_stage(self, L, n, m, blk, off, omp, doE, p, thr2, P, iterations=0, nest=None):
        """Stages E1-E7 of the module docstring for a batch of cut sides: returns the device
        addresses of the Ritz vectors U_E (n x p per problem, leading dimension n), their Ritz values
        (descending, d_e at offsets oS) and the number of directions above the threshold (d_cnt).
        L is the leading dimension of the correlation matrix; blk / off / omp are device addresses
        of the diagonal block, the off-diagonal block and the rows of Omega on the other side."""
        torch = self.torch
        el = self.elem
        ncs = len(n)

        def offsets(sizes):
            o = np.concatenate(([0], np.cumsum(sizes)))
            return o[:-1], int(o[-1])

        oY, tY = offsets(n * p)
        oB, tB = offsets(m * p)
        oR, tR = offsets(p * p)
        oS, tS = offsets(p)
        d_Y, d_U0, d_W1 = self._alloc(tY), self._alloc(tY), self._alloc(tY)
        d_Bt, d_Q2 = self._alloc(tB), self._alloc(tB)
        d_R, d_Z, d_T, d_X = self._alloc(tR), self._alloc(tR), self._alloc(tR), self._alloc(tR)
        d_sig = self._alloc(tS, real=True, zero=True)
        d_e = self._alloc(tS, real=True, zero=True)
        d_cnt = torch.zeros(ncs, dtype=torch.int32, device=self.device)
        d_scr = self._alloc(ncs * P * PANEL_W)
        Yp, U0p, W1p = (t.data_ptr() + oY * el for t in (d_Y, d_U0, d_W1))
        Btp, Q2p = (t.data_ptr() + oB * el for t in (d_Bt, d_Q2))
        Rp, Zp, Tp, Xp = (t.data_ptr() + oR * el for t in (d_R, d_Z, d_T, d_X))
        sigp, ep = d_sig.data_ptr() + oS * 8, d_e.data_ptr() + oS * 8
        cntp = d_cnt.data_ptr() + np.arange(ncs) * 4
        scrp = d_scr.data_ptr() + np.arange(ncs) * P * PANEL_W * el
        zero = np.zeros(ncs, np.int64)
        ld1 = np.maximum(n, 1)

        # E1: Y = F Omega (nest = (x, side, C, Omega): running sums over the nested blocks)
        if nest is not None:
            self.nested_products("F", L, nest[2], nest[3], L, nest[0], nest[1], Yp, p, ld1)
        else:
            self.gemm(0, 1.0, 0.0, off, omp, Yp, n, p, m, L, L, ld1)
        # E2: Q = qr(Y)
        slab_rows = 1024 if self.dtype == nat.TMF_C128 else 2048        # row limit of the slab kernel (registers)
        house_ok = self.range_qr == "house" and int(np.max(p[doE], initial=0)) <= 64
        house = house_ok and int(np.max(m[doE], initial=0)) <= slab_rows    # the longer slabs: F^H Q is m x p

        def _rqr(ptr, rows_):
            """Orthonormal factor of every slab; returns the addresses it lives at afterwards."""
            if house_ok and int(np.max(rows_[doE], initial=0)) <= slab_rows:
                # Q stays in the buffer the kernel builds it in (no copy back over the slab)
                ptr = np.array(ptr, np.int64)
                ptr[doE] = self.house_slab(ptr[doE], rows_[doE], rows_[doE], p[doE], inplace=False)
            elif self.range_qr == "house":
                self.house_general(np.asarray(ptr, np.int64)[doE], rows_[doE], rows_[doE], p[doE])
            else:
                self.bcgs2(ptr[doE], rows_[doE], rows_[doE], zero[doE], p[doE], scrp[doE], passes=3)
            return ptr

        Yp = _rqr(Yp, n)
        # E3: B^H = F^H Q  (m x p), R = Q2^H B^H.  One round of orthogonal (subspace) iteration first:
        # the range-finder error of a direction is ~ sigma_(p+1) / sigma_i, which is only 5e-6 for the
        # weakest kept mode when the spectrum decays slowly (measured: random BdG chain, L = 512);
        # Y <- F orth(F^H Q) cubes that ratio.
        for it in range(iterations + 1):
            self.gemm(1, 1.0, 0.0, off, Yp, Btp, m, p, n, L, ld1, np.maximum(m, 1))
            if house and iterations == 0:
                # only R^H = (F^H Q)^H Q2 is needed below: factor B^H in place and take R^H from the kernel - no Q2,
                # no copy, no product (R is unique up to the phases of its rows, which the Jacobi does not see)
                d_R.zero_()
                self.house_slab(Btp[doE], m[doE], np.maximum(m, 1)[doE], p[doE], r_only=Rp[doE], r_ld=np.maximum(p, 1)[doE])
                break
            d_Q2.copy_(d_Bt)
            Q2p = _rqr(d_Q2.data_ptr() + oB * el, m)
            if it < iterations:
                self.gemm(0, 1.0, 0.0, off, Q2p, Yp, n, p, m, L, np.maximum(m, 1), ld1)
                Yp = _rqr(Yp, n)
        # R^H = (F^H Q)^H Q2 (lower triangular; its columns are graded by the singular values, which is
        # the form one-sided Jacobi diagonalises in few sweeps: 11.2 ms -> measured below for R itself)
        if not (house and iterations == 0):
            self.gemm(1, 1.0, 0.0, Btp, Q2p, Rp, p, p, m, np.maximum(m, 1), np.maximum(m, 1), np.maximum(p, 1))
        # E4: Jacobi SVD of R^H: left singular vectors Z (= right ones of R), sigma; columns below the
        # threshold zeroed
        self.jacobi(Rp, Zp, sigp, cntp, thr2, p, np.maximum(p, 1), np.maximum(p, 1), left_only=True)
        # E5: U0 = Q Z
        self.gemm(0, 1.0, 0.0, Yp, Zp, U0p, n, p, p, ld1, np.maximum(p, 1), ld1)
        # E6: T = U0^H (A U0), Jacobi eigen-decomposition
        self.gemm(0, 1.0, 0.0, blk, U0p, W1p, n, p, n, L, ld1, ld1)
        self.gemm(1, 1.0, 0.0, U0p, W1p, Tp, p, p, n, ld1, ld1, np.maximum(p, 1))
        self.jacobi(Tp, Xp, ep, 0, 0.0, p, np.maximum(p, 1), np.maximum(p, 1))
        # E7: U_E = U0 X  (reuses the Y buffer; Q is no longer needed)
        self.gemm(0, 1.0, 0.0, U0p, Xp, Yp, n, p, p, ld1, np.maximum(p, 1), ld1)
        UEp = Yp

        return dict(UEp=UEp, oS=oS, d_e=d_e, d_cnt=d_cnt, d_sig=d_sig, ld1=ld1,
                    keep=(d_Y, d_U0, d_W1, d_Bt, d_Q2, d_R, d_Z, d_T, d_X, d_sig, d_scr))

    range_floor_tol = 1e-11
    range_qr = os.environ.get("TMF_RANGE_QR", "house")      # "house": LDS-panel Householder; "bcgs": blocked Gram-Schmidt
    filled_cholqr = os.environ.get("TMF_FILLED_CHOLQR", "1") == "1"   # panel method of the filled-basis QR
    det_method = os.environ.get("TMF_DET_METHOD", "ppt")              # "ppt" | "reduced" (A/B switch)
    filled_passes = int(os.environ.get("TMF_FILLED_PASSES", "1"))     # projection passes of the filled-basis QR
    host_threads = int(os.environ.get("TMF_HOST_THREADS", min(32, os.cpu_count() or 1)))   # enumeration / site preparation
    filled_blocks = int(os.environ.get("TMF_FILLED_BLOCKS", "64"))     # outer block width of the filled-basis Gram-Schmidt (64 | 16)
    # The two Gram-Schmidt chains of the filled bases (left / right blocks) on two streams: 0.3 ms faster device-resident,
    # but with a tensor download in flight the second kernel stream ended up behind the 27 ms copy (73 instead of 30 ms
    # per conversion, measured) - the runtime multiplexes streams onto few hardware queues.  Off unless asked for.
    one_stream = os.environ.get("TMF_ONE_STREAM", "1") == "1"
    filled_fused = os.environ.get("TMF_FILLED_FUSED", "1") == "1"       # 64-column blocks of the filled bases in one launch (A/B: 0)
    # "local": pivoting inside the 64 x 64 diagonal blocks, everything else MFMA GEMMs, fully pivoted fallback when a pivot
    # is small | "blocked": fully pivoted, multi-launch | "single": one workgroup per site | "fallback": local, then
    # always the fallback (tests)
    lu_method = os.environ.get("TMF_LU", "local")

    def _fetch_async(self, tensors):
        """Asynchronous device -> pinned host copies of small result tensors; returns (wait handle, NumPy
        views).  The views are valid once the handle is ready and until the next call with the same slots."""
        cache = getattr(self, "_fetch_cache", None)
        if cache is None:
            cache = self._fetch_cache = {}
        views = []
        for slot, t in enumerate(tensors):
            key = (slot, t.dtype)
            h = cache.get(key)
            if h is None or h.numel() < t.numel():
                h = cache[key] = self.torch.empty(max(int(t.numel() * 1.5), 1024), dtype=t.dtype, pin_memory=True)
            h[: t.numel()].copy_(t.reshape(-1), non_blocking=True)
            views.append(h[: t.numel()].numpy())
        return _GpuWait(self.torch, self.device), views

    def _sweeps_tensor(self):
        """Largest sweep count of every Jacobi launch since the last call, as one small device tensor."""
        t = (self.torch.stack(self._sweep_max) if self._sweep_max
             else self.torch.zeros(1, dtype=self.torch.int32, device=self.device))
        self._sweep_max = []
        return t

    def _hbuf(self, name, shape, dtype, zero=False, pinned=False):
        """Persistent host scratch (grow-only, pre-faulted).  Fresh ``np.zeros`` arrays of this size
        are lazily mapped, and their first-touch page faults inside the 16 enumeration threads
        serialise on the process's mmap lock (measured: 110 ms instead of 26 ms for the L = 1024
        enumeration).  Contents are undefined unless ``zero`` is set; nothing handed to the caller
        may alias these buffers."""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        pool = getattr(self, "_hpool", None)
        if pool is None:
            pool = self._hpool = {}
        raw = pool.get(name)
        if raw is None or raw.nbytes < nbytes:
            if pinned:   # page-locked: the async upload needs no staging copy (kept alive next to the view)
                t_raw = self.torch.empty(max(int(nbytes * 1.25), 4096), dtype=self.torch.uint8, pin_memory=True)
                raw = t_raw.numpy()
                pool[name + "/tensor"] = t_raw
            else:
                raw = np.empty(max(int(nbytes * 1.25), 4096), np.uint8)
            raw.fill(0)
            pool[name] = raw
        out = raw[:nbytes].view(dt).reshape(shape)
        if zero:
            out.fill(0)
        return out

    def default_sink(self):
        if self._sink is None:
            self._sink = PinnedSink(self.torch)
        return self._sink

    def _gmax(self, values, err=None):
        """Elementwise maximum over all ranks of a sharded conversion (identity on one rank): decisions that
        change the orbitals of a cut - range-finder width, subspace iteration - must be the same on both
        ranks that hold a shard-boundary cut, or their copies of it differ by a gauge.  ``err``: an exception this
        rank has run into since the last meeting point; it is raised here - on EVERY rank (the others get
        ``multi_gpu.RankFailure``) - so that no rank waits in a collective the failing one never enters."""
        from .multi_gpu import collective_max
        return collective_max(self.coord, values, err)

    def _finish(self, mps):
        mps.info = {"range_finder_iterations": self.range_iterations_used, "range_finder_columns": self.range_width,
                    "range_finder_smallest_sigma": self.range_floor,
                    "checks": dict(getattr(self, "check_results", {}))}
        return mps

    range_ladder = (P_RANGE, 128, 256)   # widths of the range finder tried in turn

    def entangled_stage_adaptive(self, *args):
        """Blocking form of :meth:`entangled_stage_adaptive_gen`."""
        return _drive(self.entangled_stage_adaptive_gen(*args))

    def entangled_stage_adaptive_gen(self, L, n, m, blk, off, doE, thr2, cs_b, x, side, Cp):
        """Runs the entangled stage with the narrowest adequate range finder (generator: yields a wait
        handle wherever results have to come down from the GPU).

        Adequacy is CHECKED, not assumed.  A direction with singular value s_i is found with angle error
        ~ s_P / s_i (s_P: smallest singular value captured by the P columns) and enters the state with
        weight ~ s_i, so the state error is ~ s_P for every direction.  If s_P exceeds
        ``self.range_floor_tol`` (1e-11: two orders below the 1e-9 parity tolerance on Schmidt values;
        measured on the L=1024 headline case: s_P = 1.6e-13, |dS| vs oracle 1e-12; the Nambu engine
        uses the rounding floor because its pairing construction needs the orbitals themselves) the
        stage is repeated with one round of subspace iteration, which cubes the ratio.  If that is
        still not enough, or a cut has P or more directions above the threshold, the next width of
        ``range_ladder`` is tried (p > 64 runs the block Jacobi kernels in global memory); beyond the
        ladder the call raises instead of returning degraded orbitals.
        x, side: cut positions in matrix indices and block side; Cp: the column-major matrix.
        Returns the stage dict plus P, p and host copies of the Ritz values / counts."""
        el = self.elem
        reason = ""
        for P in self.range_ladder:
            d_Om = self._alloc(L * P)
            nat.check(self.lib.tmf_fill_normal(self.dtype, d_Om.data_ptr(), L * P, 0x5EED1, self.stream), "fill")
            omp = d_Om.data_ptr() + np.where(side == 0, x, 0) * el      # rows of Omega on the other side
            p = np.where(doE, np.minimum(P, np.minimum(n, m)), 0)
            nest = None if Cp is None else (x, side, Cp, d_Om.data_ptr())   # (None: cut sides of several matrices)
            full = doE & (p == P) & (P < np.minimum(n, m))              # cuts the range finder truncates
            st = self.entangled_stage(L, n, m, blk, off, omp, doE, p, thr2, P, iterations=0, nest=nest)
            w_, (h_sig, h_cnt, h_e, h_sw) = self._fetch_async([st["d_sig"], st["d_cnt"], st["d_e"], self._sweeps_tensor()])
            yield w_
            nat.check_jacobi_sweeps(h_sw, "Jacobi SVD / eigendecomposition of a cut (replaces eigh, slater.py:347)")
            h_sig, h_cnt, h_e, oS = h_sig.copy(), h_cnt.copy(), h_e.copy(), st["oS"]
            worst = max((h_sig[oS[i] + P - 1] for i in np.nonzero(full)[0]), default=0.0)
            sat = np.nonzero(full & (h_cnt >= p))[0]
            # the decisions below are taken on the maximum over ALL ranks of a sharded conversion (_gmax)
            worst, any_sat = self._gmax([worst, float(sat.size > 0)])
            self.range_floor = float(worst)
            its = 0
            if worst > self.range_floor_tol:
                st = self.entangled_stage(L, n, m, blk, off, omp, doE, p, thr2, P, iterations=1, nest=nest)
                its = 1
                w_, (h_sig, h_cnt, h_e, h_sw) = self._fetch_async([st["d_sig"], st["d_cnt"], st["d_e"], self._sweeps_tensor()])
                yield w_
                nat.check_jacobi_sweeps(h_sw, "Jacobi SVD / eigendecomposition of a cut (replaces eigh, slater.py:347)")
                h_sig, h_cnt, h_e, oS = h_sig.copy(), h_cnt.copy(), h_e.copy(), st["oS"]
                bad = [i for i in np.nonzero(full)[0] if h_sig[oS[i] + P - 1] > 4.6e-4 * thr2**0.5]
                sat = np.nonzero(full & (h_cnt >= p))[0]
                if bad:   # (s_P / sqrt(thr2))^3 > 1e-10 even after the iteration
                    reason = (f"cut {cs_b[bad[0]]}: smallest captured singular value "
                              f"{h_sig[oS[bad[0]] + P - 1]:.1e} vs threshold {thr2 ** 0.5:.1e} with {P} columns")
                any_bad, any_sat = self._gmax([float(len(bad) > 0), float(sat.size > 0)])
                if any_bad:
                    reason = reason or "another rank's cut needs a wider range finder"
                    continue
            if any_sat:
                reason = (f"cut {cs_b[sat[0]]}: {P} or more orbitals above the range-finder threshold" if sat.size
                          else "another rank's cut has more orbitals above the threshold than the range finder is wide")
                continue
            st.update(P=P, p=p, d_Om=d_Om, range_iterations=its, h_e=h_e, h_cnt=h_cnt)
            self.range_iterations_used, self.range_width = its, P
            return st
        raise NotImplementedError(f"entanglement rank beyond the widest range finder ({self.range_ladder[-1]} "
                                  f"columns): {reason}")

    # ------------------------------------------------------------------ H -> C on the device
    def negative_projector(self, H, tol=1e-14, max_iter=100):
        """Projector onto the negative eigenspace of a Hermitian matrix, C = (1 - sign(H)) / 2, by the
        Newton-Schulz iteration  X <- X (3 - X^2) / 2  from X = H / ||H||: the occupied-orbital projector
        ``v v^H`` of ``correlation_matrix`` (slater.py:1150-1180, pfaffian.py:302-393) without an
        eigensolver - two MFMA GEMMs per step, ~1.44 log2(||H|| / gap) + 5 steps.  Convergence
        (max |1 - X^2| <= tol, evaluated by tmf_recon_error_batched) is checked every other step.
        Returns (C as a host array, number of steps)."""
        torch = self.torch
        H = np.asarray(H)
        n = len(H)
        assert H.shape == (n, n), f"Got non-square {H.shape} Hamiltonian"
        cplx = np.iscomplexobj(H)
        self.dtype = nat.TMF_C128 if cplx else nat.TMF_F64
        self.elem = 16 if cplx else 8
        torch.cuda.current_stream(self.device).synchronize()
        self._pin_off = 0
        Hc = np.ascontiguousarray(H, np.complex128 if cplx else np.float64)
        scale = min(np.linalg.norm(Hc), np.abs(Hc).sum(axis=0).max())     # >= spectral norm
        if not scale > 0:
            raise ValueError("zero Hamiltonian: every orbital is a zero mode")
        # row-major storage of H^T is column-major storage of H
        d_X = torch.from_numpy(np.ascontiguousarray(Hc.T / scale).reshape(-1)).to(self.device)
        d_T, d_Y = self._alloc(n * n), self._alloc(n * n)
        one = [n]
        steps, dev = 0, np.inf
        while steps < max_iter:
            self.gemm(0, 1.0, 0.0, [d_X.data_ptr()], [d_X.data_ptr()], [d_T.data_ptr()], one, one, one, one, one, one)
            if steps % 2 == 0:
                d_dev = self.recon_errors([dict(T=0, X=d_X.data_ptr(), Y=d_X.data_ptr(), w=None, rows=n, cols=n, q=0,
                                                inner=n, ldx=n, ldy=n, mode=1)])
                dev = float(d_dev.cpu().numpy()[0])
                if dev <= tol:
                    break
                if not np.isfinite(dev):
                    raise FloatingPointError("Newton-Schulz iteration diverged")
            d_Y.copy_(d_X)
            self.gemm(0, -0.5, 1.5, [d_X.data_ptr()], [d_T.data_ptr()], [d_Y.data_ptr()], one, one, one, one, one, one)
            d_X, d_Y = d_Y, d_X
            steps += 1
        else:
            raise RuntimeError(f"sign iteration did not converge in {max_iter} steps (max |1 - X^2| = {dev:.1e}): "
                               "the Hamiltonian has (near-)zero modes")
        X = d_X.cpu().numpy().reshape(n, n).T      # column-major on the device
        self._keep.clear()
        C = 0.5 * (np.eye(n) - X)
        return (C + C.conj().T) / 2, steps

    def lowest_projector(self, H, N, max_bisections=80):
        """Projector onto the N lowest eigenvectors of a Hermitian matrix (``v[:, :N] v[:, :N]^H`` of slater.py:1174-1179)
        without an eigensolver: bisection on the chemical potential mu, the number of levels below mu counted as the trace
        of :meth:`negative_projector` of H - mu (converged loosely while searching, to 1e-14 at the end).  A degenerate
        level at the Fermi energy (levels N and N + 1 equal) has no unique projector: ValueError, where LAPACK's ordering
        decides in the reference.  Returns (C, mu, sign iterations in total)."""
        H = np.asarray(H)
        n = len(H)
        assert H.shape == (n, n), f"Got non-square {H.shape} Hamiltonian"
        if not 0 <= N <= n:
            raise ValueError(f"cannot occupy {N} of {n} orbitals")
        if N == 0 or N == n:
            return (np.zeros_like(H) if N == 0 else np.eye(n, dtype=H.dtype)), None, 0
        bound = float(min(np.linalg.norm(H), np.abs(H).sum(axis=0).max()))      # >= spectral radius
        lo, hi, total = -1.001 * bound - 1e-300, 1.001 * bound + 1e-300, 0     # fewer than N levels below lo, at least N below hi
        eye = np.eye(n, dtype=H.dtype)
        mu, shift = 0.5 * (lo + hi), 0.0
        for _ in range(max_bisections):
            try:
                C, steps = self.negative_projector(H - (mu + shift) * eye, tol=1e-7, max_iter=70)
            except RuntimeError:          # mu sits on a level: move a little inside the bracket
                shift = (hi - lo) * (0.0137 if shift == 0.0 else -1.7 * shift / (hi - lo))
                if abs(shift) > 0.4 * (hi - lo):
                    break
                continue
            total += steps
            cnt = int(np.rint(np.trace(C).real))
            if cnt == N:
                C, steps = self.negative_projector(H - (mu + shift) * eye)
                return C, mu + shift, total + steps
            if cnt < N:
                lo = mu + shift
            else:
                hi = mu + shift
            mu, shift = 0.5 * (lo + hi), 0.0
            if hi - lo <= 1e-13 * bound:
                break
        raise ValueError(f"no gap between level {N} and level {N + 1} (chemical potential bracket [{lo}, {hi}]): "
                         f"{N} particles do not fill a shell, the ground state is not a unique Slater determinant")

    # ------------------------------------------------------------------ the sweep
    sweep_impl = os.environ.get("TMF_SWEEP", "cpp")    # "cpp": tmf_sweep_* (csrc/sweep.cpp); "python": run_gen below (A/B)

    def run(self, C, trunc, ortho_center, unit_cell_width, threads=None, download=True, site_range=None, sink=None):
        """One C -> MPS conversion.  The sweep itself - descriptor construction, stage sequencing, the two host round
        trips - runs in C++ behind the staged C ABI ``tmf_sweep_*``; this method checks arguments, applies the adaptive
        range-finder rule (reduced over the ranks of a sharded conversion) and wraps the result.

        C            (L, L) NumPy array, or a row-major torch tensor already resident in HBM
        download     True: tensors in host memory on return.  "async": returns once everything is enqueued; the tensors
                     land in page-locked host memory while the caller goes on (the next conversion's kernels overlap
                     the 1.5 GB transfer); ``result.wait()`` - called by the site objects on first access - blocks
                     until they are there.  False keeps the tensors in HBM and returns no site blocks
        site_range   (a, b): only sites a <= i < b and the cuts next to them (one rank's shard)
        sink         where the host copy of the result lives (default: page-locked memory; the multi-GPU path passes
                     shared-memory segments, multi_gpu.ShmSink)"""
        if self.sweep_impl != "cpp":
            return _drive(self.run_gen(C, trunc, ortho_center, unit_cell_width, threads, download, site_range, sink=sink))
        import ctypes
        torch, lib = self.torch, self.lib
        t_all = time.perf_counter()
        if self._ctx is None:
            ctx = ctypes.c_void_p()
            nat.check(lib.tmf_ctx_create(self.device.index or 0, ctypes.byref(ctx)), "tmf_ctx_create")
            self._ctx = ctx
            import weakref
            self._ctx_finalizer = weakref.finalize(self, lib.tmf_ctx_destroy, ctypes.c_void_p(ctx.value))
            self._ctx_finalizer.atexit = False      # at interpreter exit the process gives everything back anyway
        ctx = self._ctx
        self._tickets = [f for f in self._tickets if not self._ticket_done(f[0])]
        flags = 0
        if isinstance(C, torch.Tensor):
            torch.cuda.current_stream(self.device).synchronize()     # the sweep runs on the context's own streams
            d_Crm = C.reshape(-1)
            cplx = d_Crm.is_complex()
            L = int(round(d_Crm.numel() ** 0.5))
            c_ptr, flags, keep_c = d_Crm.data_ptr(), nat.SWEEP_C_ON_DEVICE, d_Crm
        else:
            C = np.asarray(C)
            cplx = np.iscomplexobj(C)
            C = np.ascontiguousarray(C, np.complex128 if cplx else np.float64)
            L = len(C)
            c_ptr, keep_c = C.ctypes.data, C
        self.dtype, self.elem = (nat.TMF_C128, 16) if cplx else (nat.TMF_F64, 8)
        s_lo, s_hi = site_range if site_range is not None else (0, L)
        sectors = _sector_list(trunc, L)
        sec = None if sectors is None else np.ascontiguousarray(sectors, np.int64)
        flags |= (nat.SWEEP_CHECKS if self.checks else 0) | (nat.SWEEP_TIME_KERNELS if self.time_gemm else 0)
        flags |= (nat.SWEEP_RANGE_BCGS if self.range_qr != "house" else 0) | (0 if self.filled_cholqr else nat.SWEEP_NO_CHOLQR)
        flags |= nat.SWEEP_TWO_PASSES if self.filled_passes >= 2 else 0
        if self.lu_method not in ("local", "blocked", "single", "fallback"):
            raise ValueError(f"lu_method {self.lu_method!r}: expected 'local', 'blocked', 'single' or 'fallback'")
        flags |= {"single": nat.SWEEP_LU_SINGLE, "blocked": nat.SWEEP_LU_PIVOTED, "fallback": nat.SWEEP_LU_FORCE_FALLBACK}.get(
            self.lu_method, 0)
        flags |= nat.SWEEP_NARROW_BCGS if self.filled_blocks == 16 else 0
        flags |= nat.SWEEP_ONE_STREAM if self.one_stream else 0
        flags |= 0 if self.filled_fused else nat.SWEEP_UNFUSED_BCGS
        flags |= nat.SWEEP_DET_DIRECT if self.force_direct_det else (nat.SWEEP_DET_REDUCED if self.det_method != "ppt" else 0)
        par = nat.SweepParams(L=L, chi_max=int(trunc.chi_max or 0), svd_min=float(trunc.svd_min),
                              degeneracy_tol=float(trunc.degeneracy_tol), sectors=None if sec is None else sec.ctypes.data,
                              ortho_center=int(ortho_center), site_lo=int(s_lo), site_hi=int(s_hi),
                              n_sectors=0 if sec is None else int(sec.size), is_complex=int(cplx),
                              host_threads=int(threads or self.host_threads), flags=flags)
        # ---- entangled stage with the narrowest adequate range finder (see entangled_stage_adaptive_gen): adequacy is
        # CHECKED, the decisions are taken on the maximum over all ranks.  An exception of this rank (bad arguments, out of
        # memory, the Jacobi sweep cap) is held until the next reduction and raised there on all ranks (_gmax) ----
        worst, sat, weak, sw, bad = (ctypes.c_double(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64())

        def attempt(fn, *args):
            try:
                fn(*args)
            except Exception as exc:
                return exc
            return None

        def stage(P, its):
            nat.check(lib.tmf_sweep_entangled(ctx, P, its, ctypes.byref(worst), ctypes.byref(sat), ctypes.byref(weak),
                                              ctypes.byref(sw), ctypes.byref(bad)), "tmf_sweep_entangled")
            nat.check_jacobi_sweeps([sw.value], "Jacobi SVD / eigendecomposition of a cut (replaces eigh, slater.py:347)")

        err = attempt(lambda: nat.check(lib.tmf_sweep_begin(ctx, c_ptr, ctypes.byref(par)), "tmf_sweep_begin"))
        reason, accepted = "", False
        for P in self.range_ladder:
            err = err or attempt(stage, P, 0)
            g_worst, any_sat = self._gmax([worst.value, float(sat.value)], err)
            self.range_floor, its = float(g_worst), 0
            if g_worst > self.range_floor_tol:
                err = attempt(stage, P, 1)
                its = 1
                any_weak, any_sat = self._gmax([float(weak.value), float(sat.value)], err)
                if any_weak:
                    reason = (f"cut {bad.value}: smallest captured singular value {worst.value:.1e} vs threshold "
                              f"{(trunc.svd_min ** 2 * (1 - trunc.svd_min ** 2)) ** 0.5:.1e} with {P} columns" if weak.value
                              else "another rank's cut needs a wider range finder")
                    continue
            if any_sat:
                reason = (f"cut {bad.value}: {P} or more orbitals above the range-finder threshold" if sat.value
                          else "another rank's cut has more orbitals above the threshold than the range finder is wide")
                continue
            self.range_iterations_used, self.range_width, accepted = its, P, True
            break
        if not accepted:
            raise NotImplementedError(f"entanglement rank beyond the widest range finder ({self.range_ladder[-1]} "
                                      f"columns): {reason}")

        del keep_c      # (a device-resident C has been read by now: the entangled stage synchronised the launch stream)
        dims = nat.SweepDims()
        nat.check(lib.tmf_sweep_sites(ctx, ctypes.byref(dims)), "tmf_sweep_sites")
        want_out = download is not False
        entries, total = ShardArrays.plan(nat.sweep_spec(dims, cplx, want_out))
        buf, keep = (sink or self.default_sink()).alloc(total)
        shard = ShardArrays.create(buf, entries, dict(L=int(L), s_lo=int(s_lo), s_hi=int(s_hi), ortho_center=int(ortho_center),
                                                      complex=bool(cplx)), keepalive=keep)
        ptrs = nat.SweepPtrs(**{k_: shard.arrays[k_].ctypes.data for k_ in nat.SWEEP_ARRAYS})
        ticket = ctypes.c_int64()
        nat.check(lib.tmf_sweep_download(ctx, ctypes.byref(ptrs), int(want_out), ctypes.byref(ticket)), "tmf_sweep_download")
        self._tickets.append((ticket.value, keep))      # the host buffer stays referenced until the copy is done
        chk_names = ["vL is not unitary", "vL does not diagonalise C_LL", "vR is not unitary", "vR does not diagonalise C_RR",
                     "vL and vR do not SVD C_LR"]
        state = {}

        def wait():
            if "checks" in state:
                return
            vals, nchk = (ctypes.c_double * 8)(), ctypes.c_int32()
            nat.check(lib.tmf_sweep_wait(ctx, ticket.value, vals, ctypes.byref(nchk)), "tmf_sweep_wait")
            state["checks"] = dict(zip(chk_names, (float(v) for v in vals[: nchk.value])))
            shard.wait = None
            shard.check_det()

        shard.wait = wait
        self.timings = self._stage_timings()
        res = MPSData.from_shards([shard], ortho_center, unit_cell_width, self.timings, with_sites=want_out)
        if want_out and isinstance(keep, torch.Tensor):
            # the page-locked array all blocks are views of, as a tensor (gutzwiller: re-upload in one copy)
            o_ = entries["out"][2]
            res._flat_t = keep[o_: o_ + int(dims.out_elems) * self.elem].view(torch.complex128 if cplx else torch.float64)
        if download == "async":
            self.check_results = {}
            res._lazy_checks = lambda: (wait(), state["checks"])[1]
            return self._finish(res)
        wait()
        self.check_results = state["checks"]
        self.timings = self._stage_timings()
        self.timings["total"] = time.perf_counter() - t_all
        res.timings = dict(self.timings)
        return self._finish(res)

    def _ticket_done(self, ticket):
        r = self.lib.tmf_sweep_query(self._ctx, ticket)
        if r < 0:                       # a HIP error of the download, not "done"
            nat.check(r, "tmf_sweep_query")
        return r != 0

    def close(self):
        """Releases the C++ sweep context: its two device memory sets, output blocks, page-locked staging memory and streams
        (GBs at benchmark size).  Called by a finalizer when the engine is garbage-collected."""
        fin = getattr(self, "_ctx_finalizer", None)
        if fin is not None:
            fin()
            self._ctx_finalizer = None
        self._ctx = None

    def _stage_timings(self):
        """Host wall time per stage of the last sweep (seconds), and - with ``time_gemm`` - the kernel times measured with
        HIP events on the launch stream (``self.kernel_info``: gemm_ms / gemm_flops over all MFMA GEMM launches of the
        conversion, det_ms / det_flops of the dominant determinant launch, det_all_ms, n_det)."""
        import ctypes
        info = nat.SweepInfo()
        nat.check(self.lib.tmf_sweep_info_get(self._ctx, ctypes.byref(info)), "tmf_sweep_info_get")
        self.kernel_info = info
        out = {}
        for i in range(16):
            name = self.lib.tmf_sweep_stage_name(i).decode()
            if name:
                out[name] = info.stage_ms[i] * 1e-3
        return out

    def device_result(self):
        """(device address, element count) of the tensors of the last conversion (valid until the next one)."""
        import ctypes
        p, n = ctypes.c_uint64(), ctypes.c_int64()
        nat.check(self.lib.tmf_sweep_device_out(self._ctx, ctypes.byref(p), ctypes.byref(n)), "tmf_sweep_device_out")
        return p.value, n.value

    def run_gen(self, C, trunc, ortho_center, unit_cell_width, threads=None, download=True, site_range=None,
                diag=None, presynced=False, sink=None, second=None):
        """One C -> MPS conversion as a generator: yields a wait handle (``ready()`` / ``block()``) at the
        three places where the host has to wait - eigenvalues coming down, the enumeration thread, the
        final results - so that several site ranges can be interleaved (:func:`run_pipelined`).

        C            (L, L) NumPy array, or a row-major torch tensor already resident in HBM
        download     True: tensors in host memory on return.  "async": the call returns once everything is
                     enqueued; the tensors land in page-locked host memory while the caller goes on (the next
                     conversion's kernels overlap the 1.5 GB transfer); ``result.wait()`` - called by the site
                     objects on first access - blocks until they are there.  False keeps the tensors in HBM
                     (``self.d_out``) and returns no site blocks
        sink         where the host copy of the result lives (default: page-locked memory; the multi-GPU path
                     passes shared-memory segments, multi_gpu.ShmSink)
        site_range   (a, b): only sites a <= i < b and the cuts next to them (one rank's shard)
        second       a second chain in the same sweep and tensors between cuts of the two (``slater.C_to_iMPS``,
                     slater.py:1499-1553): ``{"C": (L, L) matrix, "oc": its centre cut, "cross": [...]}``.  A cross site is
                     ``dict(mode=0|1, phys=bool, bra=(chain, cut), ket=(chain, cut), skip=rows of the bra block that come
                     before the orbitals it shares with the ket)``; they follow the regular sites in the result (site
                     numbers s_hi, s_hi + 1, ...), cuts of chain 1 have the bond numbers L + 1 + cut.  ``site_range`` may
                     then be empty
        """
        torch = self.torch
        t_all = time.perf_counter()
        self.timings = {}
        self.det_events = []
        self.gemm_events = []
        if not presynced:
            torch.cuda.current_stream(self.device).synchronize()  # staging arena of the previous call is free
        self._pin_off = 0
        self._inflight = [f for f in self._inflight if not f[0].query()]
        if isinstance(C, torch.Tensor):
            d_Crm = C.reshape(-1)
            cplx = d_Crm.is_complex()
            L = int(round(d_Crm.numel() ** 0.5))
            if diag is None:
                diag = torch.real(d_Crm[:: L + 1]).cpu().numpy().astype(np.float64)
        else:
            C = np.asarray(C)
            cplx = np.iscomplexobj(C)
            C = np.ascontiguousarray(C, np.complex128 if cplx else np.float64)
            L = len(C)
            if diag is None:
                diag = np.real(np.diagonal(C)).astype(np.float64)
            d_Crm = None
        self.dtype = nat.TMF_C128 if cplx else nat.TMF_F64
        self.elem = 16 if cplx else 8
        oc = ortho_center
        el = self.elem
        cutoff = trunc.svd_min**2  # slater.py:318
        thr2 = cutoff * (1.0 - cutoff)
        threads = threads or min(16, os.cpu_count() or 1)
        s_lo, s_hi = site_range if site_range is not None else (0, L)

        t0 = time.perf_counter()
        if d_Crm is None:
            d_Crm = torch.from_numpy(C.reshape(-1)).to(self.device)
        d_C = self._alloc(L * L)
        nat.check(self.lib.tmf_transpose(self.dtype, d_Crm.data_ptr(), d_C.data_ptr(), L, self.stream), "tmf_transpose")
        Cp = d_C.data_ptr()
        Cps, diags, ocs = [Cp], [diag], [oc]
        cross = []
        if second is not None:
            C2 = np.ascontiguousarray(second["C"], np.complex128 if cplx else np.float64)
            assert C2.shape == (L, L), "the second chain is embedded in a matrix of the same size"
            d_C2rm = torch.from_numpy(C2.reshape(-1)).to(self.device)
            d_C2 = self._alloc(L * L)
            nat.check(self.lib.tmf_transpose(self.dtype, d_C2rm.data_ptr(), d_C2.data_ptr(), L, self.stream), "tmf_transpose")
            Cps.append(d_C2.data_ptr()), diags.append(np.real(np.diagonal(C2)).astype(np.float64)), ocs.append(int(second["oc"]))
            cross = list(second["cross"])
        KEY = L + 1                                   # bond number of cut b of chain c: c * KEY + b
        self._tick("upload", t0)

        # ---- cut-side problems --------------------------------------------------------------
        need = set()                                  # (chain, cut, side)
        for i in range(s_lo, s_hi):
            sd_ = 0 if i < oc else 1
            need.add((0, i, sd_)), need.add((0, i + 1, sd_))
        for x_ in cross:
            sd_ = int(x_["mode"]) & 1
            need.add((x_["bra"][0], x_["bra"][1], sd_)), need.add((x_["ket"][0], x_["ket"][1], sd_))
        for c_, oc_ in enumerate(ocs):
            if (c_, oc_, 1) in need or (c_, oc_, 0) in need or (c_ == 0 and s_lo < s_hi and s_lo <= oc_ <= s_hi):
                need.add((c_, oc_, 0)), need.add((c_, oc_, 1))  # the centre's right orbitals are paired with its left ones
        trip = sorted(need)
        cs_mat = np.array([t_[0] for t_ in trip], np.int64)
        cs_b = np.array([t_[1] for t_ in trip], np.int64)
        cs_side = np.array([t_[2] for t_ in trip], np.int64)
        cs_key = cs_mat * KEY + cs_b
        ncs = len(cs_b)
        n = np.where(cs_side == 0, cs_b, L - cs_b)
        m = L - n
        Cbase = np.array(Cps, np.int64)[cs_mat]
        blk = Cbase + np.where(cs_side == 0, 0, (cs_b + cs_b * L)) * el      # A = C_LL or C_RR
        off = Cbase + np.where(cs_side == 0, cs_b * L, cs_b) * el            # F (n x m)
        # the right side of a centre cut is paired with the left side through C_RL (block_svd, slater.py:407)
        doE = (n > 0) & (m > 0)
        centres = []                                  # (left problem, right problem) of every chain's centre cut
        for c_, oc_ in enumerate(ocs):
            if (c_, oc_, 0) in need:
                cl_ = int(np.nonzero((cs_mat == c_) & (cs_b == oc_) & (cs_side == 0))[0][0])
                cr_ = int(np.nonzero((cs_mat == c_) & (cs_b == oc_) & (cs_side == 1))[0][0])
                centres.append((cl_, cr_))
                doE[cr_] = False
        has_centre = bool(centres)
        centre_L, centre_R = centres[0] if centres else (-1, -1)
        # sites: the regular ones of chain 0, then the cross sites; mode bit 1 = no physical leg
        st_mode = [0 if i < oc else 1 for i in range(s_lo, s_hi)] + [int(x_["mode"]) & 1 for x_ in cross]
        st_phys = [True] * (s_hi - s_lo) + [bool(x_["phys"]) for x_ in cross]
        st_bkey = [(i if i < oc else i + 1) for i in range(s_lo, s_hi)] + [x_["bra"][0] * KEY + x_["bra"][1] for x_ in cross]
        st_kkey = [(i + 1 if i < oc else i) for i in range(s_lo, s_hi)] + [x_["ket"][0] * KEY + x_["ket"][1] for x_ in cross]
        st_skip = [0] * (s_hi - s_lo) + [int(x_.get("skip", 0)) for x_ in cross]
        st_mode, st_phys, st_bkey, st_kkey, st_skip = (np.array(v, np.int64) for v in (st_mode, st_phys, st_bkey, st_kkey, st_skip))

        def offsets(sizes):
            o = np.concatenate(([0], np.cumsum(sizes)))
            return o[:-1], int(o[-1])

        t0 = time.perf_counter()
        st = yield from self.entangled_stage_adaptive_gen(L, n, m, blk, off, doE, thr2, cs_b, cs_b, cs_side,
                                                          Cp if second is None else None)
        P, p = st["P"], st["p"]
        UEp, oS, ld1 = st["UEp"], st["oS"], st["ld1"]
        self._tick("E_entangled", t0)

        # ---- host round trip 1: eigenvalues -> classification, filled counts ------------------
        t0 = time.perf_counter()
        h_e, h_cnt = st["h_e"], st["h_cnt"]
        csums = np.stack([np.concatenate(([0.0], np.cumsum(d_))) for d_ in diags])
        n_fermions = np.round(csums[:, -1]).astype(np.int64)  # slater.py:414
        n_fermion = int(n_fermions[0])
        tr = np.where(cs_side == 0, csums[cs_mat, cs_b], csums[cs_mat, -1] - csums[cs_mat, cs_b])
        # kept Ritz values per cut side as a padded 2-D array (descending inside each row)
        colP = np.arange(P)
        valid = colP[None, :] < h_cnt[:, None]
        E2 = np.where(valid, h_e[np.minimum(oS[:, None] + colP[None, :], max(len(h_e) - 1, 0))], 0.0)
        x_hi = np.sum(valid & (E2 >= 1 - cutoff), axis=1)   # kept but 'filled' by slater.py:350
        x_lo = np.sum(valid & (E2 < cutoff), axis=1)
        ent0 = np.where(doE, x_hi, 0)                        # first entangled column inside U_E
        k = np.where(doE, h_cnt - x_hi - x_lo, 0).astype(np.int64)
        e_side = [E2[i, a_: b_] for i, (a_, b_) in enumerate(zip(ent0.tolist(), (ent0 + k).tolist()))]  # views
        esum = np.where((colP[None, :] >= ent0[:, None]) & (colP[None, :] < (ent0 + k)[:, None]), E2, 0.0).sum(axis=1)
        # centre: right-side eigenvalues are 1 - e_L reversed (slater.py:386 convention)
        for cl_, cr_ in centres:
            k[cr_] = k[cl_]
            e_side[cr_] = (1.0 - e_side[cl_])[::-1].copy()
            esum[cr_] = e_side[cr_].sum()
        nf = np.clip(np.round(tr - esum).astype(np.int64), 0, n - k)
        self._tick("host_classify", t0)

        def host_phase():
            # ---- host: best-first enumeration for every cut (one threaded C++ call) ----------------
            t0 = time.perf_counter()
            sectors = _sector_list(trunc, L)
            # lookup tables: cut-side problem of (cut, side), position of a cut in this rank's list (-1 = absent)
            nkey = KEY * len(Cps)
            side_idx = np.full((nkey, 2), -1, np.int64)
            side_idx[cs_key, cs_side] = np.arange(ncs)
            my_cuts = np.unique(cs_key)
            ncut = len(my_cuts)
            cpos = np.full(nkey, -1, np.int64)
            cpos[my_cuts] = np.arange(ncut)
            iL, iR = side_idx[my_cuts, 0], side_idx[my_cuts, 1]
            hasL, hasR = iL >= 0, iR >= 0
            src = np.where(hasL, iL, iR)                       # the side whose eigenvalues define e_left
            kk_cut = k[src].astype(np.int32)
            nf_src = nf[src]
            other = n_fermions[my_cuts // KEY] - kk_cut - nf_src   # slater.py:167 / :172
            nfl = np.where(hasL, nf_src, np.where(hasR, other, 0)).astype(np.int32)
            nfr = np.where(hasL, np.where(hasR, nf[np.maximum(iR, 0)], other), nf_src).astype(np.int32)
            # left eigenvalues of every cut, flat: from the left block as they are, from the right block as
            # 1 - e reversed (slater.py:386); e_side[i] are the k[i] kept Ritz values of problem i
            e_off = (np.cumsum(kk_cut) - kk_cut).astype(np.int64)
            tot = int(kk_cut.sum())
            owner = np.repeat(np.arange(ncut), kk_cut)
            t_in = np.arange(tot) - e_off[owner]
            rev = ~hasL[owner]
            rows_ = src[owner]
            vals = E2[rows_, ent0[rows_] + np.where(rev, kk_cut[owner] - 1 - t_in, t_in)]   # kept Ritz values
            e_pool = np.concatenate((np.where(rev, 1.0 - vals, vals), np.zeros(1)))
            sec_arr = None if sectors is None else np.ascontiguousarray(sectors, np.int64)
            cap = int(trunc.chi_max) + 1 if trunc.chi_max else 4096
            self.timings["host_enum_setup"] = time.perf_counter() - t0
            while True:
                c_sets = self._hbuf("c_sets", (ncut, cap, 2), np.uint64)
                c_lam = self._hbuf("c_lam", (ncut, cap), np.float64)
                c_q = self._hbuf("c_q", (ncut, cap), np.int32)
                c_chi, c_chk = np.zeros(ncut, np.int64), np.zeros(ncut, np.int64)
                st = self.lib.tmf_cut_vectors_batch(
                    ncut, nat._p(e_pool), nat._p(e_off), nat._p(kk_cut), nat._p(nfl), int(trunc.chi_max or 0),
                    float(trunc.svd_min), float(trunc.degeneracy_tol), None if sec_arr is None else nat._p(sec_arr),
                    0 if sec_arr is None else sec_arr.size, cap, nat._p(c_sets), nat._p(c_lam), nat._p(c_q), nat._p(c_chi),
                    nat._p(c_chk), threads)
                if st == -3 and not trunc.chi_max and int(c_chi.max()) > cap:
                    cap = int(c_chi.max()) + 1  # unlimited chi: grow the per-cut capacity and redo
                    continue
                nat.check(st, "tmf_cut_vectors_batch")
                break
            self.timings["host_enum_native"] = time.perf_counter() - t0 - self.timings["host_enum_setup"]
            if np.any(c_chi == 0):
                raise ValueError("No Schmidt vectors left after filtering by `trunc_par.sectors`!")  # slater.py:668
            if logger.isEnabledFor(logging.INFO):
                for j, b in enumerate(my_cuts):
                    logger.info("bond %d: %d Schmidt modes, checked %d subsets, kept %d, norm %.12g", b,
                                kk_cut[j], c_chk[j], int(c_chi[j]),
                                float(np.sqrt(np.dot(c_lam[j, :int(c_chi[j])], c_lam[j, :int(c_chi[j])]))))
            # enumeration outputs of this rank's cuts (scratch, reused by the next sweep: the result object
            # gets copies, see the end of run_gen)
            bond_arrays = dict(my_cuts=my_cuts.astype(np.int64), c_sets=c_sets, c_lam=c_lam, c_q=c_q, c_chi=c_chi, c_chk=c_chk,
                               e_pool=e_pool, e_off=e_off, kk_cut=kk_cut, nfl=nfl, nfr=nfr)

            self.timings["host_enumerate"] = time.perf_counter() - t0

            # ---- host: per-site integer preparation (one threaded C++ call) --------------------------
            t0 = time.perf_counter()
            ns = len(st_mode)
            my_sites = s_lo + np.arange(ns)
            mode = st_mode.astype(np.int32)
            bb, kb_ = st_bkey, st_kkey
            ib, ik = side_idx[bb, mode], side_idx[kb_, mode]
            cb_i, ck_i = cpos[bb], cpos[kb_]
            chi_b, chi_k = c_chi[cb_i], c_chi[ck_i]
            jobs = np.zeros(ns, nat.site_job)
            jobs["mode"], jobs["cut_b"], jobs["cut_k"] = mode | np.where(st_phys != 0, 0, 2).astype(np.int32), cb_i, ck_i
            jobs["k_b"], jobs["nf_b"], jobs["k_k"], jobs["nf_k"] = k[ib], nf[ib], k[ik], nf[ik]
            mb_cap = k[ib] + nf[ib] + 1
            mk_cap = np.maximum(k[ik] + nf[ik], 1)
            sec_cap = k[ik] + 2
            n_bound = np.minimum(255, k[ik] + np.maximum(0, nf[ik] + k[ik] - nf[ib]) + 1)
            idx_cap = (2 * chi_b + chi_k) * n_bound + 16
            jobs["sec_cap"] = sec_cap
            jobs["row_off"], rs_tot = offsets(mb_cap)
            jobs["col_off"], cs_tot = offsets(mk_cap)
            jobs["bra_off"], br_tot = offsets(2 * chi_b)
            jobs["sec_off"], sc_tot = offsets(sec_cap)
            jobs["idx_off"], ix_tot = offsets(idx_cap)
            jobs["idx_cap"] = idx_cap
            row_sel, row_sign = np.zeros(rs_tot + 1, np.int32), np.zeros(rs_tot + 1, np.int8)
            col_sel, col_sign = np.zeros(cs_tot + 1, np.int32), np.zeros(cs_tot + 1, np.int8)
            # bra_p / bra_alpha end up in the returned SiteData -> fresh arrays, faulted in here
            bra_p, bra_alpha = np.empty(br_tot + 1, np.int32), np.empty(br_tot + 1, np.int32)
            bra_p.fill(0), bra_alpha.fill(0)
            sec_buf = np.zeros(sc_tot + 1, nat.sector)
            pool = self._hbuf("idx_pool", (ix_tot + 1,), np.uint8, pinned=True)   # uploaded straight from here
            souts = np.zeros(ns, nat.site_out)
            t1 = time.perf_counter()
            nat.check(self.lib.tmf_site_prepare_batch(
                ns, nat._p(jobs), nat._p(c_sets), nat._p(c_q), nat._p(c_chi), cap, nat._p(row_sel), nat._p(row_sign),
                nat._p(col_sel), nat._p(col_sign), nat._p(bra_p), nat._p(bra_alpha), nat._p(sec_buf), nat._p(pool),
                nat._p(souts), threads), "tmf_site_prepare_batch")
            self.timings["host_site_native"] = time.perf_counter() - t1
            self.timings["host_site_prepare"] = time.perf_counter() - t0

            return dict(bond_arrays=bond_arrays, jobs=jobs, souts=souts, row_sel=row_sel, row_sign=row_sign, col_sel=col_sel,
                        col_sign=col_sign, bra_p=bra_p, bra_alpha=bra_alpha, sec_buf=sec_buf, pool=pool, mode=mode,
                        ib=ib, ik=ik, chi_b=chi_b, chi_k=chi_k, my_sites=my_sites, ns=ns)

        # the integer host phase (C++, GIL released) runs while the filled-basis launches are enqueued
        import threading
        hp = {}

        def _run_host():
            try:
                hp["out"] = host_phase()
            except BaseException as exc:  # re-raised on the main thread
                hp["exc"] = exc

        th = threading.Thread(target=_run_host)
        th.start()

        # ---- F: orbital matrices V = [U_E (k) | Q_f (nf)] -------------------------------------
        t0 = time.perf_counter()
        ncolV = k + nf
        oV, tV = offsets(n * ncolV)
        d_V = self._alloc(tV)
        Vp = d_V.data_ptr() + oV * el
        # entangled columns (renormalised copy); centre-right: C_RL U_E(left), reversed, odd columns flipped
        # canonical gauge of the entangled Ritz vectors (tmf_gauge_desc; csrc/sweep.cpp filled_stage)
        gsel = np.nonzero(doE & (k > 0))[0]
        if gsel.size:
            gdesc = np.zeros(gsel.size, nat.gauge_desc)
            starts = []
            for t_, i_ in enumerate(gsel):
                e_ = np.asarray(e_side[i_], np.float64)
                st_ = np.arange(len(e_), dtype=np.int32)
                for j_ in range(1, len(e_)):
                    w_ = min(min(e_[j_], 1.0 - e_[j_]), min(e_[j_ - 1], 1.0 - e_[j_ - 1]))
                    if not abs(e_[j_] - e_[j_ - 1]) > 1e-13 + 1e-9 * w_:
                        st_[j_] = st_[j_ - 1]
                starts.append(st_)
            t_st = self._up(np.concatenate(starts))
            so_ = np.concatenate(([0], np.cumsum([len(x_) for x_ in starts])))[:-1]
            gdesc["V"] = (UEp + ent0 * ld1 * el)[gsel]
            gdesc["start"] = t_st.data_ptr() + so_ * 4
            gdesc["n"], gdesc["k"], gdesc["ld"], gdesc["from_top"] = n[gsel], k[gsel], ld1[gsel], (cs_side[gsel] == 1)
            t_gd = self._up(gdesc)
            nat.check(self.lib.tmf_canonical_gauge_batched(self.dtype, t_gd.data_ptr(), int(gsel.size), int(n[gsel].max()),
                                                           int(k[gsel].max()), self.stream),
                      "tmf_canonical_gauge_batched")
        cp = doE.copy()
        for centre_L, centre_R in centres:
            perm = _group_order(e_side[centre_L], trunc.degeneracy_tol) if doE[centre_L] else None
            if perm is not None:
                # left orbitals of the centre cut inside a group of eigenvalues closer than degeneracy_tol: the SVD the
                # reference takes of v_L^H C_LR v_R per group (utils.py:66-94) sorts them by descending sqrt(e (1 - e)); e stays
                cp[centre_L] = False
                i_ = centre_L
                src0 = UEp[i_] + ent0[i_] * ld1[i_] * el
                one = np.ones(len(perm), np.int64)
                self.colcopy(src0 + perm * ld1[i_] * el, Vp[i_] + np.arange(len(perm)) * ld1[i_] * el, n[i_] * one, one,
                             ld1[i_] * one, ld1[i_] * one)
        self.colcopy((UEp + ent0 * ld1 * el)[cp], Vp[cp], n[cp], k[cp], ld1[cp], ld1[cp])
        for centre_L, centre_R in centres:
            if not (k[centre_L] > 0 and n[centre_R] > 0):
                continue
            d_pair = self._alloc(n[centre_R] * k[centre_L])
            self.gemm(0, 1.0, 0.0, [off[centre_R]], [Vp[centre_L]], [d_pair.data_ptr()], [n[centre_R]], [k[centre_L]],
                      [m[centre_R]], [L], [ld1[centre_L]], [ld1[centre_R]])
            # The partners of weak orbitals (sigma -> 1e-6) carry errors ~1e-7 from the division by sigma
            # (measured: |V_R^H V_R - 1| = 3.7e-7 on spinful chains), so they are Gram-Schmidt
            # orthonormalised in order of DECREASING sigma: strong partners stay as computed, weak ones
            # are corrected against them.  Column permutations are k x k (signed) permutation GEMMs.
            kc, nR, ldR = int(k[centre_L]), int(n[centre_R]), int(ld1[centre_R])
            eL_ = e_side[centre_L]
            order = np.argsort(-(eL_ * (1.0 - eL_)), kind="stable")
            cdt = np.complex128 if cplx else np.float64
            Pm, Sm = np.zeros((kc, kc), cdt), np.zeros((kc, kc), cdt)
            for a_, j_ in enumerate(order):
                Pm[j_, a_] = 1.0                                        # T[:, a] = pair[:, order[a]]
                Sm[a_, kc - 1 - j_] = -1.0 if (kc - 1 - j_) & 1 else 1.0  # slater.py:410 reversal + signs
            t_P = self._up(np.ascontiguousarray(Pm.T).reshape(-1))       # column-major upload
            t_S = self._up(np.ascontiguousarray(Sm.T).reshape(-1))
            d_T = self._alloc(nR * kc)
            self.gemm(0, 1.0, 0.0, [d_pair.data_ptr()], [t_P.data_ptr()], [d_T.data_ptr()], [nR], [kc], [kc], [ldR],
                      [kc], [ldR])
            d_scrc = self._alloc((kc + 1) * PANEL_W)
            self.bcgs2(np.array([d_T.data_ptr()]), np.array([nR]), np.array([ldR]), np.array([0]), np.array([kc]),
                       np.array([d_scrc.data_ptr()]))
            self.gemm(0, 1.0, 0.0, [d_T.data_ptr()], [t_S.data_ptr()], [Vp[centre_R]], [nR], [kc], [kc], [ldR], [kc],
                      [ldR])
        centre_L, centre_R = centres[0] if centres else (-1, -1)
        # filled: Y = A Omega_f, projected off U_E and orthonormalised below
        maxnf = int(nf.max()) if ncs else 0
        if maxnf > 0:
            d_OmF = self._alloc(L * maxnf)
            nat.check(self.lib.tmf_fill_normal(self.dtype, d_OmF.data_ptr(), L * maxnf, 0xF111ED, self.stream), "fill")
            Vf = Vp + k * ld1 * el
            # one multiplication by A: the filled space has eigenvalue >= 1 - 1e-12, everything that is
            # not projected off with U_E below has eigenvalue <= 1e-12 (the reference's own cutoff)
            if second is None:
                self.nested_products("A", L, Cp, d_OmF.data_ptr(), L, cs_b, cs_side, Vf, nf, ld1)
            else:     # (two matrices: plain products, rows of Omega by global index as in the nested form)
                self.gemm(0, 1.0, 0.0, blk, d_OmF.data_ptr() + np.where(cs_side == 1, cs_b, 0) * el, Vf, n, nf, n, L, L, ld1)
            d_scr2 = self._alloc(int((ncolV.max() + 1) * PANEL_W) * ncs)
            scr2 = d_scr2.data_ptr() + np.arange(ncs) * int((ncolV.max() + 1) * PANEL_W) * el
            has = nf > 0
            # One projection pass per block: the columns are random combinations inside the filled space, so
            # a block's residual after the projection keeps >= (n_f - j) / n_f of its norm (no cancellation
            # for a second pass to repair: measured |V^H V - 1| <= 1e-14 at the centre cut either way); the
            # Cholesky-QR inside the block runs twice.
            self.bcgs2(Vp[has], n[has], ld1[has], k[has], ncolV[has], scr2[has], cholqr=self.filled_cholqr,
                       passes=self.filled_passes)
        # self-check of the centre cut (testing.py:131-177; slater.py:419-420 runs it only there)
        chk_names, d_chk = [], None
        items = []
        for iL, iR in (centres if self.checks else []):
            if not doE[iL]:
                continue
            qL, qR, kc = int(ncolV[iL]), int(ncolV[iR]), int(k[iL])
            wL = np.concatenate((e_side[iL], np.ones(int(nf[iL]))))
            wR = np.concatenate((e_side[iR], np.ones(int(nf[iR]))))
            eL = e_side[iL]
            sv = np.sqrt(eL * (1.0 - eL)) * (-1.0) ** (np.arange(kc)[::-1])  # slater.py:266-268
            mk = lambda **kw: kw  # noqa: E731
            items += [
                mk(T=0, X=Vp[iL], Y=Vp[iL], w=None, rows=qL, cols=qL, q=0, inner=int(n[iL]), ldx=int(ld1[iL]),
                   ldy=int(ld1[iL]), mode=1),
                mk(T=blk[iL], X=Vp[iL], Y=Vp[iL], w=wL, rows=int(n[iL]), cols=int(n[iL]), q=qL, ldt=L,
                   ldx=int(ld1[iL]), ldy=int(ld1[iL]), mode=0),
                mk(T=0, X=Vp[iR], Y=Vp[iR], w=None, rows=qR, cols=qR, q=0, inner=int(n[iR]), ldx=int(ld1[iR]),
                   ldy=int(ld1[iR]), mode=1),
                mk(T=blk[iR], X=Vp[iR], Y=Vp[iR], w=wR, rows=int(n[iR]), cols=int(n[iR]), q=qR, ldt=L,
                   ldx=int(ld1[iR]), ldy=int(ld1[iR]), mode=0),
                mk(T=off[iL], X=Vp[iL], Y=Vp[iR], w=sv, rows=int(n[iL]), cols=int(n[iR]), q=kc, ldt=L,
                   ldx=int(ld1[iL]), ldy=int(ld1[iR]), mode=0, y_reverse=1),
            ]
        if items:
            chk_names = ["vL is not unitary", "vL does not diagonalise C_LL", "vR is not unitary",
                         "vR does not diagonalise C_RR", "vL and vR do not SVD C_LR"]
            if len(items) > 5:    # (second chain: the same five deviations of its centre cut)
                chk_names = chk_names + [nm + " (second chain)" for nm in chk_names]
            d_chk = self.recon_errors(items)
        self._tick("F_filled", t0)

        # ---- S1 ahead of the host results: O = V_bra^H V_ket needs the orbital matrices only ---------
        t0 = time.perf_counter()
        e_mode = st_mode
        e_sidx = np.full((KEY * len(Cps), 2), -1, np.int64)
        e_sidx[cs_key, cs_side] = np.arange(ncs)
        e_ib = e_sidx[st_bkey, e_mode]
        e_ik = e_sidx[st_kkey, e_mode]
        cb, ck = ncolV[e_ib], ncolV[e_ik]           # columns of V_bra / V_ket
        nb_rows = n[e_ib] - st_skip                  # contraction length = bra orbitals (shared with the ket)
        oO, tO = offsets(cb * ck)
        d_O = self._alloc(tO)
        Op = d_O.data_ptr() + oO * el
        # right mode: the physical orbital is row 0 of the ket block (none between two bases of the same orbitals)
        Vk_sub = Vp[e_ik] + np.where((e_mode == 1) & (st_phys != 0), 1, 0) * el
        physp = Vp[e_ik] + np.where(e_mode == 1, 0, nb_rows) * el
        self.gemm(1, 1.0, 0.0, Vp[e_ib] + st_skip * el, Vk_sub, Op, cb, ck, nb_rows, ld1[e_ib], ld1[e_ik], np.maximum(cb, 1))
        self._tick("S_overlap_gemm", t0)

        yield _ThreadWait(th)
        th.join()
        if "exc" in hp:
            raise hp["exc"]
        ho = hp["out"]
        jobs, souts, pool, mode = ho["jobs"], ho["souts"], ho["pool"], ho["mode"]
        row_sel, row_sign, col_sel, col_sign = ho["row_sel"], ho["row_sign"], ho["col_sel"], ho["col_sign"]
        bra_p, bra_alpha, sec_buf = ho["bra_p"], ho["bra_alpha"], ho["sec_buf"]
        ib, ik, chi_b, chi_k, my_sites, ns = ho["ib"], ho["ik"], ho["chi_b"], ho["chi_k"], ho["my_sites"], ho["ns"]
        L_all, L = L, ns  # from here on "L" counts the sites of this shard
        # the index pool (19 MB) was written into pinned memory by the site preparation; it goes up on a side
        # stream while the overlap GEMM and the LU run (on the launch stream the copy sat between the LU and
        # the determinant kernel: 1.5 ms of idle SIMDs)
        t_pin = self._hpool["idx_pool/tensor"][: pool.nbytes]
        t_pool = torch.empty(pool.nbytes, dtype=torch.uint8, device=self.device)
        self._keep.append(t_pool)
        self._up_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._up_stream):
            t_pool.copy_(t_pin, non_blocking=True)
            pool_ready = torch.cuda.Event()
            pool_ready.record(self._up_stream)
        pool_upload = (t_pool, pool_ready)

        # ---- S1/S2: overlaps and W assembly ---------------------------------------------------
        t0 = time.perf_counter()
        mb, mk, ka = (souts[f].astype(np.int64) for f in ("mb", "mk", "k_always"))
        sbv, skv = souts["sb"].astype(np.int64), souts["sk"].astype(np.int64)
        assert np.array_equal(ib, e_ib) and np.array_equal(ik, e_ik)
        oW, tW = offsets(mb * mk)
        d_W = self._alloc(tW)
        d_det = self._alloc(L)
        Wp = d_W.data_ptr() + oW * el
        detp = d_det.data_ptr() + np.arange(L) * el
        # selection arrays (already flat, at the offsets given to the C++ call)
        rs_off, cs_off = jobs["row_off"], jobs["col_off"]
        t_rs, t_cs, t_rg, t_cg = self._up(row_sel), self._up(col_sel), self._up(row_sign), self._up(col_sign)
        gd = np.zeros(L, nat.gather_desc)
        gd["src"], gd["dst"] = Op, Wp
        gd["row_sel"], gd["col_sel"] = t_rs.data_ptr() + rs_off * 4, t_cs.data_ptr() + cs_off * 4
        gd["row_sign"], gd["col_sign"] = t_rg.data_ptr() + rs_off, t_cg.data_ptr() + cs_off
        gd["phys"] = physp
        gd["rows"], gd["cols"] = mb, mk
        gd["lds_"], gd["ldd"], gd["ldp"] = np.maximum(cb, 1), np.maximum(mb, 1), ld1[ik]
        t_gd = self._up(gd)
        nat.check(self.lib.tmf_gather_signed_batched(self.dtype, t_gd.data_ptr(), L, self.stream), "gather")
        # S3: det_always + Schur complement
        sd = np.zeros(L, nat.schur_desc)
        sd["W"], sd["S"], sd["det"] = Wp, 0, detp
        sd["mb"], sd["mk"], sd["k"], sd["ldw"], sd["lds"] = mb, mk, ka, np.maximum(mb, 1), 1
        t_sd = self._up(sd)
        nat.check(self.lib.tmf_lu_schur_batched(self.dtype, t_sd.data_ptr(), L, int(mb.max()), self.stream), "lu_schur")
        self._tick("S_overlap_schur", t0)

        # ---- S4: all minors ----------------------------------------------------------------------
        t0 = time.perf_counter()
        out_off, out_tot = offsets(souts["out_elems"])
        t_pool, pool_ready = pool_upload
        d_out = self._alloc(out_tot)
        Sp = Wp + (ka + ka * np.maximum(mb, 1)) * el
        n_det = 0
        torch.cuda.current_stream(self.device).wait_event(pool_ready)   # the determinant kernels read the pool
        flop_per_det = (8.0 / 3.0) if cplx else (2.0 / 3.0)  # LU of an n x n complex / real matrix
        nsec = souts["n_sectors"].astype(np.int64)
        sec_ptr = np.concatenate(([0], np.cumsum(nsec)))
        rest_keys = None
        if self.det_method == "ppt":
            # tiles of the pivoted-exchange kernel for every sector it takes, built by the host library
            ds = np.zeros(L, nat.det_site)
            ds["S"], ds["scale"], ds["lds"] = Sp, detp, np.maximum(mb, 1)
            ds["idx_base"] = t_pool.data_ptr() + jobs["idx_off"]
            ds["out_base"] = d_out.data_ptr() + out_off * el
            n_rest, lds_max, fl3, npairs = (np.zeros(1, np.int64), np.zeros(1, np.int32), np.zeros(1), np.zeros(1, np.int64))
            args = (L, nat._p(jobs), nat._p(souts), nat._p(sec_buf), nat._p(ds), el, 16384)
            nt = int(self.lib.tmf_det_tiles_build(*args, None, 0, None, 0, nat._p(n_rest), nat._p(lds_max), nat._p(fl3),
                                                  nat._p(npairs)))
            tiles = np.zeros(max(nt, 1), nat.det_desc)
            rest_keys = np.zeros(max(int(n_rest[0]), 1), np.int64)
            nt = int(self.lib.tmf_det_tiles_build(*args, nat._p(tiles), nt, nat._p(rest_keys), len(rest_keys), nat._p(n_rest),
                                                  nat._p(lds_max), nat._p(fl3), nat._p(npairs)))
            rest_keys = rest_keys[: int(n_rest[0])]
            if nt > 0:
                t_dd = self._up(tiles[:nt])
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(self.device))
                widest = int(max(souts["sb"].max(initial=0), souts["sk"].max(initial=0)))    # as csrc/sweep.cpp: mask width of the launch
                if int((sec_buf["c1"] - sec_buf["c0"]).max(initial=0)) > 2048:
                    widest = max(widest, 33)
                nat.check(self.lib.tmf_det_ppt_batched_w(self.dtype, t_dd.data_ptr(), nt, int(lds_max[0]), 32 if widest <= 32 else 64,
                                                         self.stream), "tmf_det_ppt_batched")
                ev1.record(torch.cuda.current_stream(self.device))
                self.det_events.append(("ppt", ev0, ev1, float(fl3[0]) * flop_per_det, int(npairs[0])))
                n_det += int(npairs[0])
        # general path: every sector (TMF_DET_METHOD=reduced) or the ones the exchange kernel does not take
        if rest_keys is None:
            sec_site = np.repeat(np.arange(L), nsec)
            sec_loc = np.arange(int(nsec.sum())) - sec_ptr[:-1][sec_site]
        else:
            sec_site, sec_loc = (rest_keys >> 32).astype(np.int64), (rest_keys & 0xFFFFFFFF).astype(np.int64)
        sec_all = sec_buf[jobs["sec_off"][sec_site] + sec_loc]
        if len(sec_all):
            nq = sec_all["n"].astype(np.int64)
            nsb_ = (sec_all["r1"] - sec_all["r0"]).astype(np.int64)
            nsk_ = (sec_all["c1"] - sec_all["c0"]).astype(np.int64)
            cls_ = np.where(nq <= 32, nq, 64)  # exact order for n <= 32 (templated kernels), generic above
            ta = np.clip(_cdiv(4096, nsk_), 1, nsb_)
            ntile = _cdiv(nsb_, ta)
            tsec = np.repeat(np.arange(len(sec_all)), ntile)
            tloc = np.arange(int(ntile.sum())) - np.repeat(np.cumsum(ntile) - ntile, ntile)
            tsite = sec_site[tsec]
            dd_all = np.zeros(len(tsec), nat.det_desc)
            dd_all["S"], dd_all["scale"] = Sp[tsite], detp[tsite]
            pbase = t_pool.data_ptr() + jobs["idx_off"][tsite]
            dd_all["bra_idx"] = pbase + sec_all["bra_off"][tsec]
            dd_all["ket_idx"] = pbase + sec_all["ket_off"][tsec]
            dd_all["out"] = d_out.data_ptr() + (out_off[tsite] + sec_all["out_off"][tsec]) * el
            dd_all["sb"], dd_all["sk"], dd_all["lds"] = sbv[tsite], skv[tsite], np.maximum(mb[tsite], 1)
            dd_all["n"], dd_all["nsb"], dd_all["nsk"] = nq[tsec], nsb_[tsec], nsk_[tsec]
            dd_all["a0"] = tloc * ta[tsec]
            dd_all["a1"] = np.minimum(nsb_[tsec], dd_all["a0"] + ta[tsec])
            a16 = lambda x: (x + 15) & ~15  # noqa: E731
            tcls = cls_[tsec]
            # LDS per workgroup (det_gather.hip): M, index lists, then per wave the gathered rows
            # M[rows(a), :] (n*sk) + 64 scratch elements; class 64 instead holds one n x n minor
            gpw = np.where(nq[tsec] <= 8, 8, np.where(nq[tsec] <= 16, 4, 2))
            lneed = (a16(sbv[tsite] * skv[tsite] * el) + a16(nsk_[tsec] * nq[tsec]) + a16(ta[tsec] * nq[tsec])
                     + np.where(tcls == 64, nq[tsec] ** 2 * el,
                                4 * ((nq[tsec] | 1) * skv[tsite] + gpw * (nq[tsec] + 1)) * el))
            pairs = (dd_all["a1"] - dd_all["a0"]).astype(np.int64) * dd_all["nsk"]
            # reduced-minor kernel (one Gauss-Jordan per bra row-set) whenever the sometimes-matrix has
            # <= 64 columns and 1 <= n <= 32; the direct kernel covers the rest
            use_red = (tcls >= 1) & (tcls <= 32) & (skv[tsite] <= 64) & (not self.force_direct_det)
            lneed_red = nat.reduced_det_lds(el, nq[tsec], sbv[tsite], skv[tsite], nsk_[tsec], ta[tsec]) - 16
            use_red &= lneed_red + 16 <= 160 * 1024
            lneed = np.where(use_red, lneed_red, lneed)
            # the sometimes-matrix (or the minor) does not fit next to the index lists: class 255 reads the matrix from global
            # memory and keeps only the minor in LDS (slow, general); TMF_DET_GLOBAL=1 sends every sector there (test switch)
            glob = (lneed > 160 * 1024) | (nq[tsec] > 64) | bool(int(os.environ.get("TMF_DET_GLOBAL", "0")))
            tcls = np.where(glob, 255, tcls)
            use_red = use_red & ~glob
            lneed = np.where(glob, a16(nq[tsec] ** 2 * el) + 64, lneed)
            if int(lneed.max()) > 158 * 1024:
                raise NotImplementedError(f"minors of order {int(nq[tsec][np.argmax(lneed)])} exceed the LDS of a CU")
            launches = sorted({(int(c_), bool(r_)) for c_, r_ in zip(tcls.tolist(), use_red.tolist())},
                              key=lambda cr: -int(pairs[(tcls == cr[0]) & (use_red == cr[1])].sum()))
            for cls, red in launches:
                selc = np.nonzero((tcls == cls) & (use_red == red))[0]
                if selc.size == 0:
                    continue
                # biggest tiles first
                selc = selc[np.argsort(-(pairs[selc] * (nq[tsec][selc] + 1) ** 2), kind="stable")]
                dd = dd_all[selc]
                t_dd = self._up(dd)
                flops = float((pairs[selc].astype(np.float64) * nq[tsec][selc].astype(np.float64) ** 3).sum()) * flop_per_det
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(self.device))
                fn = self.lib.tmf_det_reduced_batched if red else self.lib.tmf_det_gather_batched
                nat.check(fn(self.dtype, cls, t_dd.data_ptr(), len(dd), int(lneed[selc].max()) + 16, self.stream),
                          "tmf_det_reduced_batched" if red else "tmf_det_gather_batched")
                ev1.record(torch.cuda.current_stream(self.device))
                self.det_events.append((f"{cls}{'r' if red else ''}", ev0, ev1, flops, int(pairs[selc].sum())))
                n_det += int(pairs[selc].sum())
        self.n_det = n_det
        self._tick("S_determinants", t0)
        self.d_out = d_out  # device-resident result
        self.d_det = d_det

        # ---- result object: ONE host buffer (page-locked; a shared-memory segment on the multi-GPU path) holds the
        # bookkeeping arrays of the host phase and receives the tensors straight from the GPU ------------------
        t0 = time.perf_counter()
        cdt = np.complex128 if cplx else np.float64
        src = dict(ho["bond_arrays"])
        src.update(mode=mode.astype(np.int32), sec_off=jobs["sec_off"].astype(np.int64), nsec=nsec,
                   sectors=sec_buf, out_off=out_off.astype(np.int64), bra_off=jobs["bra_off"].astype(np.int64),
                   chi_b=chi_b.astype(np.int64), chi_k=chi_k.astype(np.int64), bra_p=bra_p, bra_alpha=bra_alpha)
        spec = {k_: (v_.dtype, v_.shape) for k_, v_ in src.items()}
        want_out = download is not False
        spec["det"] = (cdt, (L,))
        spec["out"] = (cdt, (int(out_tot) if want_out else 0,))
        entries, total = ShardArrays.plan(spec)
        buf, keep = (sink or self.default_sink()).alloc(total)
        shard = ShardArrays.create(buf, entries, dict(L=int(KEY * len(Cps) - 1), s_lo=int(s_lo), s_hi=int(s_lo + ns),
                                                      ortho_center=int(oc), complex=bool(cplx)), keepalive=keep)
        for k_, v_ in src.items():
            shard.arrays[k_][...] = v_
        self._tick("host_bonds", t0)

        # ---- host round trip 2: tensors back, on the copy stream (the launch stream is free for the next
        # conversion; the page-locked buffer and the device tensors stay referenced until the copies are done) ----
        t0 = time.perf_counter()
        cs = self._copy_stream
        cs.wait_stream(torch.cuda.current_stream(self.device))
        d_det.record_stream(cs)
        nat.check(self.lib.tmf_memcpy_async(shard.arrays["det"].ctypes.data, d_det.data_ptr(), L * el, 1, cs.cuda_stream), "D2H")
        if want_out:
            d_out.record_stream(cs)
            dst, chunk = shard.arrays["out"].ctypes.data, 64 << 20
            for o_ in range(0, int(out_tot) * el, chunk):   # in pieces: small copies of other streams get a turn
                nat.check(self.lib.tmf_memcpy_async(dst + o_, d_out.data_ptr() + o_, min(chunk, int(out_tot) * el - o_), 1,
                                                    cs.cuda_stream), "D2H")
        check_names, h_chk = chk_names, None
        if d_chk is not None:
            d_chk.record_stream(cs)
            h_chk = torch.empty(len(check_names), dtype=torch.float64, pin_memory=True)
            with torch.cuda.stream(cs):
                h_chk.copy_(d_chk[: len(check_names)], non_blocking=True)
        done = torch.cuda.Event()
        done.record(cs)
        self._inflight.append((done, keep, d_out, d_det, d_chk, h_chk))
        state = {"checked": False}

        def wait():
            if state["checked"]:
                return
            done.synchronize()
            state["checked"] = True
            shard.wait = None
            shard.check_det()

        shard.wait = wait
        self.timings["total"] = time.perf_counter() - t_all
        self._keep.clear()
        res = MPSData.from_shards([shard], oc, unit_cell_width, dict(self.timings), with_sites=want_out)
        if want_out and isinstance(keep, torch.Tensor):
            # the page-locked array all blocks are views of, as a tensor (gutzwiller: re-upload in one copy)
            o_ = entries["out"][2]
            res._flat_t = keep[o_: o_ + int(out_tot) * el].view(torch.complex128 if cplx else torch.float64)
        if download == "async":
            # the caller overlaps the download with its next conversion and calls res.wait() (the site objects do);
            # the reconstruction deviations of the centre cut are read then as well
            self.check_results = {}
            if d_chk is not None:
                def lazy_checks():
                    done.synchronize()
                    return dict(zip(check_names, (float(v) for v in h_chk.numpy())))
                res._lazy_checks = lazy_checks
            return self._finish(res)
        yield _EventWait(done)
        self.check_results = {}
        if d_chk is not None:
            self.check_results = dict(zip(check_names, (float(v) for v in h_chk.numpy())))
        wait()
        self._tick("download", t0)
        res.timings = dict(self.timings, total=time.perf_counter() - t_all)
        return self._finish(res)


def _group_order(e, tol):
    """New column order of the left orbitals of the centre cut, or None if nothing moves (sweep.cpp ``group_order``): inside
    groups of consecutive eigenvalues not further apart than ``tol`` (utils.py:71) by descending e (1 - e); values equal to
    rounding keep their order."""
    e = np.asarray(e, np.float64)
    k = len(e)
    perm = np.arange(k)
    moved = False
    a = 0
    while a < k:
        b = a
        while b + 1 < k and not abs(e[b + 1] - e[b]) > tol:
            b += 1
        for x in range(a + 1, b + 1):
            y = x
            while y > a:
                s1, s0 = e[perm[y]] * (1.0 - e[perm[y]]), e[perm[y - 1]] * (1.0 - e[perm[y - 1]])
                if not s1 > s0 * (1.0 + 1e-13):
                    break
                perm[y], perm[y - 1] = perm[y - 1], perm[y]
                moved = True
                y -= 1
        a = b + 1
    return perm if moved else None


def _sector_list(trunc, L):
    """StoppingCondition.sectors (schmidt_utils.py:67-77) as an explicit list of allowed charges."""
    s = trunc.sectors
    if s is None:
        return None
    return [q for q in range(0, 2 * L + 2) if trunc.is_sector(q)]
