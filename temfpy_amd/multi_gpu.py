"""One conversion over the GPUs of one node: contiguous site ranges, one process per GPU.

Sites are independent given C (the loops of slater.py:1301-1321 and :1326-1346 touch site i through the
cuts i and i+1 only), so rank r converts sites [a_r, b_r) and recomputes the cut on each of its range
boundaries; the kernels are deterministic and the few data-dependent decisions of the entangled stage are
taken on the maximum over all ranks (``Engine._gmax``), so both neighbours hold bit-identical copies of a
boundary cut.  Data path:

  C        rank 0 uploads it and RCCL-broadcasts the 16 MiB over xGMI (the only collective on the data path)
  tensors  every rank copies its shard through its own PCIe link into a page-locked POSIX shared-memory
           segment (``ShmSink``); the assembling process maps the segments of all ranks and builds ONE
           ``MPSData`` whose bond / site objects are views into them - no gather collective, no copy
  control  a gloo group next to the RCCL one carries the elementwise-max decisions and the (generation,
           size, busy time) triple of each rank's segment

Two ways in:

  * ranks started by a launcher (``torch.distributed.run`` or ``bench.py --gpus N``): ``ShardGroup`` on every
    rank, ``group.convert(C, ...)`` returns the assembled ``MPSData`` on rank 0;
  * ``slater.C_to_MPS(..., devices=[...])``: a ``DevicePool`` of worker processes, one per device, started
    before the calling process touches a GPU; the caller hands C over in shared memory and assembles the
    result itself.

``TMF_DRY_ENGINE=1`` replaces the GPU engine by a deterministic stand-in (``DryEngine``) so that the process
plumbing - spawn, rendezvous, segments, leases, assembly - is testable on a machine without a GPU.
"""
from __future__ import annotations

import atexit
import functools
import mmap
import os
import pickle
import socket
import subprocess
import sys
import time
import weakref

import numpy as np

from .mps_data import MPSData, ShardArrays

_LEASE_OFF = ShardArrays.HEADER_ROOM - 16      # uint64 inside the header room: 1 while a reader holds views

CHECK_NAMES = ("vL is not unitary", "vL does not diagonalise C_LL", "vR is not unitary", "vR does not diagonalise C_RR",
               "vL and vR do not SVD C_LR")      # testing.py:131-177, in the order tmf_sweep_wait reports them


class RankFailure(RuntimeError):
    """Raised on the ranks that did NOT fail when another rank of a sharded conversion did: a failure is made collective
    at the points where the ranks meet (the decisions of the entangled stage, the exchange of segment names), so every
    rank leaves the conversion together and the group stays in step for the next one."""


def collective_max(coord, values, err=None):
    """Elementwise maximum over the ranks (``coord.max``; identity without a coordinator) with a failure flag riding along:
    if any rank passes an exception, every rank raises - the failing one its own exception, the others RankFailure."""
    v = np.asarray(list(values) + [0.0 if err is None else 1.0], np.float64)
    if coord is not None:
        v = coord.max(v)
    if v[-1] > 0:
        if err is not None:
            raise err
        raise RankFailure("another rank of the sharded conversion failed")
    return v[:-1]


def shard_sites(L, oc, world):
    return list(_shard_sites(int(L), int(oc), int(world)))


@functools.lru_cache(maxsize=64)
def _shard_sites(L, oc, world):
    """Contiguous site ranges that minimise the time of the slowest rank under the cost model

        t(lo, hi) = 1.40 + sum_{i in [lo, hi)} (0.0222 - 0.0074 x_i + 0.0110 x_i^3) * 1024 / L + 1.72 max_i x_i^2   [ms],   x_i = n_i / (L / 2),

    n_i = size of the smaller block at site i.  Fitted (rms 0.16, largest residual 0.43 ms; round 3) to the HOST -> HOST time per
    conversion of 25 site ranges of the benchmark chain, conversions pipelined as ShardGroup / bench.py do - the download of
    conversion k under conversion k + 1, consecutive conversions on two alternating contexts (tools/shard_host_cost.py): a
    per-site part (host enumeration and site preparation, determinant stage ~ constant in the chi-saturated bulk, eigen / overlap
    stages ~ n^3) and a per-rank latency that grows with the largest block of the range - the per-cut kernels of a shard with
    fewer than 256 cuts take as long as their slowest workgroup, and two contexts hide about half of it.  (The round-2 fit was to
    device times of one context and off by up to 2.2 ms: the end ranks, with the most sites, are host bound.)"""
    if world > L:
        raise ValueError(f"{world} ranks for {L} sites: every rank needs at least one site")
    i = np.arange(L)
    n = np.where(i < oc, i + 1, L - i)
    x = n / max(L / 2, 1)
    c1 = np.concatenate(([0.0], np.cumsum((0.02224 - 0.00739 * x + 0.01104 * x ** 3) * (1024.0 / L))))
    x2 = x ** 2

    def greedy(c1, x2):
        c1, x2 = c1.tolist(), x2.tolist()

        def cut(T):            # every range as long as the budget T allows
            b = [0]
            while b[-1] < L and len(b) <= world:
                lo = b[-1]
                hi, mx = lo + 1, x2[lo]
                while hi < L and 1.40 + (c1[hi + 1] - c1[lo]) + 1.72 * max(mx, x2[hi]) <= T:
                    mx = max(mx, x2[hi])
                    hi += 1
                b.append(hi)
            return b

        lo_T, hi_T = 0.0, 1.40 + c1[L] + 1.72 * max(x2)
        for _ in range(50):
            mid = 0.5 * (lo_T + hi_T)
            b = cut(mid)
            if b[-1] >= L and len(b) - 1 <= world:
                hi_T = mid
            else:
                lo_T = mid
        b = cut(hi_T)
        b[-1] = L
        while len(b) - 1 < world:          # fewer ranges than ranks (tiny chains): split the longest
            k = int(np.argmax(np.diff(b)))
            b.insert(k + 1, (b[k] + b[k + 1]) // 2)
        return b

    # The greedy fill leaves the remainder to the last range: filled from the left and from the right, boundary by boundary the
    # mean of the two (a chain with its centre in the middle gets mirror-symmetric ranges).
    per_site = (0.02224 - 0.00739 * x + 0.01104 * x ** 3) * (1024.0 / L)
    left = greedy(c1, x2)
    right = greedy(np.concatenate(([0.0], np.cumsum(per_site[::-1]))), x2[::-1].copy())
    twice = [a + (L - b) for a, b in zip(left, right[::-1])]
    bounds = [m // 2 if 2 * k <= world else (m + 1) // 2 for k, m in enumerate(twice)]      # (rounded towards the nearer end)
    bounds[0], bounds[-1] = 0, L
    for r in range(1, world):          # strictly increasing: no empty range
        bounds[r] = min(max(bounds[r], bounds[r - 1] + 1), L - (world - r))
    return tuple((bounds[r], bounds[r + 1]) for r in range(world))


# ------------------------------------------------------------------------------------------------ shared memory
class Segment:
    """A POSIX shared-memory file mapped into this process (plain ``/dev/shm`` + ``mmap``: Python's
    ``multiprocessing.shared_memory`` hands attached segments to a resource tracker that unlinks them when the
    attaching process exits)."""

    def __init__(self, name, size=None, create=False):
        self.name, self.path, self.owner = name, "/dev/shm/" + name, create
        fd = os.open(self.path, os.O_RDWR | (os.O_CREAT | os.O_EXCL if create else 0), 0o600)
        try:
            if create:
                os.ftruncate(fd, size)
            self.size = os.fstat(fd).st_size
            self.map = mmap.mmap(fd, self.size)
        finally:
            os.close(fd)
        self.buf = np.frombuffer(self.map, np.uint8)
        self.registered = False

    @property
    def lease(self):
        return int(self.buf[_LEASE_OFF: _LEASE_OFF + 8].view(np.uint64)[0])

    @lease.setter
    def lease(self, v):
        self.buf[_LEASE_OFF: _LEASE_OFF + 8].view(np.uint64)[0] = v

    def unlink(self):
        if self.owner:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            self.owner = False


class ShmSink:
    """Host memory of a rank's results: shared-memory segments ``<tag>_r<rank>_g<generation>``, page-locked for
    the GPU (``tmf_host_register``) so that the tensors arrive by asynchronous DMA.  A segment is reused as soon
    as the reader has dropped the result built on it (lease word back to 0); otherwise a new generation is
    created, so results handed out earlier stay valid."""

    def __init__(self, tag, rank, lib=None):
        self.tag, self.rank, self.lib = tag, rank, lib
        self.segments = []           # generation -> Segment
        self.last = None
        atexit.register(self.close)

    def alloc(self, nbytes):
        seg = None
        for s in self.segments:
            if s.size >= nbytes and s.lease == 0:
                seg = s
                break
        if seg is None:
            size = (int(nbytes * 1.25) + (1 << 21)) & ~((1 << 21) - 1)
            seg = Segment(f"{self.tag}_r{self.rank}_g{len(self.segments)}", size, create=True)
            seg.gen = len(self.segments)
            seg.buf[: ShardArrays.HEADER_ROOM] = 0
            if self.lib is not None:
                from . import _native as nat
                nat.check(self.lib.tmf_host_register(seg.buf.ctypes.data, seg.size), "tmf_host_register")
                seg.registered = True
            self.segments.append(seg)
        self.last = seg
        return seg.buf, seg

    def close(self):
        for s in self.segments:
            if s.registered and self.lib is not None:
                self.lib.tmf_host_unregister(s.buf.ctypes.data)
                s.registered = False
            s.unlink()


class _Reader:
    """Maps the segments of all ranks in the assembling process (attachments are cached by name)."""

    def __init__(self, tag):
        self.tag, self.cache = tag, {}

    def segment(self, rank, gen):
        name = f"{self.tag}_r{rank}_g{gen}"
        seg = self.cache.get(name)
        if seg is None:
            seg = self.cache[name] = Segment(name)
        return seg

    def shard(self, rank, gen):
        """Views into the segment, and the lease tied to THEM: every array handed out (``bonds[b].masks``,
        ``sites[i].blocks`` ...) is a view whose base is the array created here, so the finalizer fires - and the rank may
        overwrite the segment - only when the result object AND every array taken from it are gone."""
        seg = self.segment(rank, gen)
        lease = np.frombuffer(seg.map, np.uint8)
        weakref.finalize(lease, _release, [seg])
        return ShardArrays.unpack(lease, keepalive=seg), seg


def assemble(reader, infos, ortho_center, unit_cell_width, timings=None):
    """``MPSData`` over the segments announced by the ranks: infos[r] = (generation, ...).  A segment stays leased
    until the returned object and every array view taken from it have been garbage-collected (``_Reader.shard``)."""
    shards = [reader.shard(r, int(info[0]))[0] for r, info in enumerate(infos)]
    shards = [s for s in shards if s.meta["s_hi"] > s.meta["s_lo"] or len(s.arrays["my_cuts"])]
    return MPSData.from_shards(shards, ortho_center, unit_cell_width, timings)


def _release(segs):
    for s in segs:
        try:
            s.lease = 0
        except (ValueError, BufferError):    # interpreter shutdown: mapping already gone
            pass


# ------------------------------------------------------------------------------------------------ rank side
class GlooMax:
    def __init__(self, dist, torch, group):
        self.dist, self.torch, self.group = dist, torch, group

    def max(self, v):
        t = self.torch.from_numpy(np.array(v, np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t.numpy()


class DryEngine:
    """Stand-in for ``engine.Engine`` without a GPU (``TMF_DRY_ENGINE=1``): fabricates the flat result of a site
    range deterministically from (L, site, bond) so that tests can follow every value through shared memory and
    the assembly, and exercises the cross-rank decision hook.  Never used by the product path."""

    def __init__(self, device=None):
        self.coord, self.timings, self.check_results = None, {}, {}
        self.range_iterations_used, self.range_width, self.range_floor = 0, 64, 0.0
        # failure injection for the tests: "rank:where:n" - rank fails in its n-th conversion (0-based) at the decision
        # point ("stage", before the collective) or after it ("sites", a failure the other ranks do not see)
        f = os.environ.get("TMF_DRY_FAIL", "")
        self._fail = tuple(f.split(":")) if f else None
        self._count = 0

    def run(self, C, trunc, ortho_center, unit_cell_width, threads=None, download=True, site_range=None, sink=None):
        from . import _native as nat
        C = np.asarray(C)
        L = len(C)
        lo, hi = site_range if site_range is not None else (0, L)
        fail = None
        if self._fail and int(self._fail[0]) == int(os.environ.get("RANK", 0)) and int(self._fail[2]) == self._count:
            fail = self._fail[1]
        self._count += 1
        # a decision that differs between the ranks unless it is reduced: each rank proposes its own range
        v = np.array([float(lo), float(hi)])
        self.decision = collective_max(self.coord, v, ValueError("injected failure before the decision") if fail == "stage" else None)
        if fail == "sites":
            raise ValueError("injected failure after the decision")
        cuts = np.arange(lo, hi + 1) if hi > lo else np.zeros(0, np.int64)
        ncut, cap, ns = len(cuts), 3, hi - lo
        src = dict(my_cuts=cuts.astype(np.int64), c_sets=np.zeros((ncut, cap, 2), np.uint64), c_lam=np.zeros((ncut, cap)),
                   c_q=np.zeros((ncut, cap), np.int32), c_chi=np.full(ncut, 2, np.int64), c_chk=np.zeros(ncut, np.int64),
                   e_pool=np.zeros(ncut + 1), e_off=np.arange(ncut, dtype=np.int64), kk_cut=np.ones(ncut, np.int32),
                   nfl=cuts.astype(np.int32), nfr=(L - cuts).astype(np.int32))
        src["c_lam"][:, 0], src["c_lam"][:, 1] = 3.0 + cuts, 4.0
        src["c_sets"][:, 1, 0] = 1
        src["e_pool"][:ncut] = 0.25 + 0.5 * cuts / (L + 1)
        sec = np.zeros(ns, nat.sector)
        sec["q"], sec["r0"], sec["r1"], sec["c0"], sec["c1"] = np.arange(lo, hi), 0, 4, 0, 2
        cdt = C.dtype if C.dtype.kind == "c" else np.float64
        src.update(mode=(np.arange(lo, hi) >= ortho_center).astype(np.int32), sec_off=np.arange(ns, dtype=np.int64),
                   nsec=np.ones(ns, np.int64), sectors=sec, out_off=8 * np.arange(ns, dtype=np.int64),
                   bra_off=4 * np.arange(ns, dtype=np.int64), chi_b=np.full(ns, 2, np.int64), chi_k=np.full(ns, 2, np.int64),
                   bra_p=np.tile(np.array([0, 0, 1, 1], np.int32), ns), bra_alpha=np.tile(np.array([0, 1, 0, 1], np.int32), ns),
                   det=np.ones(ns, cdt), out=(np.arange(8 * ns) + 8.0 * lo + C[0, 0]).astype(cdt))
        entries, total = ShardArrays.plan({k: (v_.dtype, v_.shape) for k, v_ in src.items()})
        buf, keep = sink.alloc(total)
        sh = ShardArrays.create(buf, entries, dict(L=int(L), s_lo=int(lo), s_hi=int(hi), ortho_center=int(ortho_center),
                                                   complex=bool(np.iscomplexobj(C)), decision=self.decision.tolist()),
                                keepalive=keep)
        for k, v_ in src.items():
            sh.arrays[k][...] = v_
        mps = MPSData.from_shards([sh], ortho_center, unit_cell_width, {"total": 1e-3})
        # the self-check deviations exist on the rank(s) whose range touches the centre cut (csrc/sweep.cpp: has_centre)
        checks = {nm: 1e-9 * (k + 1) for k, nm in enumerate(CHECK_NAMES)} if lo <= ortho_center <= hi else {}
        mps.info = {"checks": checks, "decision": self.decision.tolist()}
        return mps


class ShardGroup:
    """Rank-side driver of the sharded conversion.  ``torch.distributed`` must be initialised (backend "nccl" =
    RCCL, one rank per GPU; "gloo" for the one-GPU rehearsal / dry mode, where C travels as a host tensor)."""

    def __init__(self, engine, tag, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.eng, self.tag = torch, dist, engine, tag
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.data_nccl = dist.get_backend() == "nccl"
        self.ctl = dist.new_group(backend="gloo") if self.data_nccl else dist.group.WORLD
        # Consecutive conversions alternate between two engines (two contexts = two launch streams): a shard is a chain of
        # short latency-bound launches with two host round trips, and the GPU overlaps the chains of two conversions that
        # sit on different streams (tools/shard_inflight_exp.py: 2 - 9 % per conversion).  The second engine is created at
        # the second conversion; a one-off conversion never pays for it.
        self.engines, self._turn = [engine], 0
        # TMF_SHARD_FORCE_COLLECTIVES=1: run the broadcast / all-reduce / all-gather calls at world size 1 too (they are
        # identities there) - how the RCCL branch is executed on a one-GPU box (tests/test_gpu_multi.py)
        self.collective = self.world > 1 or os.environ.get("TMF_SHARD_FORCE_COLLECTIVES") == "1"
        if self.collective:
            engine.coord = GlooMax(dist, torch, self.ctl)
        self.sink = ShmSink(tag, self.rank, getattr(engine, "lib", None))
        self.reader = _Reader(tag) if self.rank == 0 else None
        self.host_threads = max(2, min(16, len(os.sched_getaffinity(0)) // max(self.world, 1)))
        self.last_busy_ms = None
        self._c_pin = None

    def convert_begin(self, C, trunc, ortho_center=None, unit_cell_width=None, wait=False):
        """Steps every rank takes: shape + C from rank 0, this rank's site range enqueued with its tensors on their way
        into the rank's segment by asynchronous DMA.  Returns a handle for :meth:`convert_end`.  Between the two calls the
        caller may begin the next conversion: its kernels overlap this one's download (as on one GPU)."""
        torch, dist = self.torch, self.dist
        t0 = time.perf_counter()
        hdr = torch.zeros(5, dtype=torch.int64)
        err0 = None
        if self.rank == 0:
            try:
                C = np.asarray(C)
                cplx = np.iscomplexobj(C)
                C = np.ascontiguousarray(C, np.complex128 if cplx else np.float64)
                if C.ndim != 2 or C.shape[0] != C.shape[1] or len(C) < self.world:
                    raise ValueError(f"correlation matrix of shape {C.shape} for {self.world} ranks")
                hdr[:4] = torch.tensor([len(C), int(cplx), ortho_center or len(C) // 2, unit_cell_width or len(C)])
            except Exception as exc:       # the other ranks wait in the broadcast: tell them instead of leaving them there
                err0, hdr[4] = exc, 1
        if self.collective:
            dist.broadcast(hdr, 0, group=self.ctl)
        if int(hdr[4]):
            raise err0 if err0 is not None else RankFailure("rank 0 could not read the correlation matrix")
        L, cplx, oc, ucw = (int(x) for x in hdr[:4])
        tdt = torch.complex128 if cplx else torch.float64
        if self.data_nccl:
            if self.rank == 0:
                # through a persistent page-locked buffer: a pageable .to(device) runs at ~6 GB/s (2.7 ms for the
                # 16 MiB every other rank waits for), a memcpy + asynchronous DMA at ~1 ms
                if self._c_pin is None or self._c_pin.numel() < L * L or self._c_pin.dtype != tdt:
                    self._c_pin = torch.empty(L * L, dtype=tdt, pin_memory=True)
                np.copyto(self._c_pin[: L * L].numpy(), C.reshape(-1))
                d_C = self._c_pin[: L * L].to(self.device, non_blocking=True)
            else:
                d_C = torch.empty(L * L, dtype=tdt, device=self.device)
            if self.collective:                             # RCCL over xGMI (complex data travels as pairs of doubles)
                dist.broadcast(torch.view_as_real(d_C) if cplx else d_C, 0)
            mat = d_C
        else:
            h_C = torch.from_numpy(C.reshape(-1)) if self.rank == 0 else torch.empty(L * L, dtype=tdt)
            if self.collective:
                dist.broadcast(torch.view_as_real(h_C) if cplx else h_C, 0)
            mat = h_C.numpy().reshape(L, L)
        rng = shard_sites(L, oc, self.world)[self.rank]
        # (the second engine only when conversions are actually pipelined: it holds its own device memory sets)
        if (not wait and self._turn >= 1 and len(self.engines) == 1 and os.environ.get("TMF_SHARD_CONTEXTS", "2") != "1"):
            second = make_engine(self.device, isinstance(self.eng, DryEngine))
            second.coord = self.eng.coord
            self.engines.append(second)
        eng = self.engines[self._turn % len(self.engines)]
        self._turn += 1
        # A failure inside the entangled stage is collective already (Engine._gmax -> collective_max): every rank arrives
        # here with an exception.  A failure after it is local; it travels with the handle to the next meeting point
        # (convert_end's all_gather), so this rank stays in step with the others until then.
        mps = seg = err = None
        try:
            mps = eng.run(mat, trunc, oc, ucw, threads=self.host_threads, download=True if wait else "async", site_range=rng,
                          sink=self.sink)
            seg = self.sink.last
            seg.lease = 1                      # taken: the next conversion gets another segment
        except Exception as exc:
            err = exc
        return dict(mps=mps, seg=seg, err=err, L=L, oc=oc, ucw=ucw, busy=(time.perf_counter() - t0) * 1e3, keep=mat)

    def convert_end(self, h):
        """Waits for the tensors of the conversion begun with handle ``h`` and assembles: the ``MPSData`` on rank 0,
        ``None`` elsewhere."""
        torch, dist = self.torch, self.dist
        t0 = time.perf_counter()
        mps_l, seg, err = h["mps"], h["seg"], h["err"]
        chk = [np.nan] * len(CHECK_NAMES)
        if err is None:
            try:
                if hasattr(mps_l, "wait"):
                    mps_l.wait()
                got = mps_l._lazy_checks() if hasattr(mps_l, "_lazy_checks") else getattr(mps_l, "info", {}).get("checks", {})
                chk = [float(got.get(nm, np.nan)) for nm in CHECK_NAMES]
            except Exception as exc:
                err = exc
        busy = h["busy"] + (time.perf_counter() - t0) * 1e3
        self.last_local = mps_l
        # payload of the meeting point: segment (generation, size), busy time, failure flag, the self-check deviations (they
        # exist only on the rank that owns the centre cut - NaN elsewhere)
        mine = torch.tensor([float(seg.gen) if seg is not None else -1.0, float(seg.size) if seg is not None else 0.0, busy,
                             0.0 if err is None else 1.0] + chk, dtype=torch.float64)
        infos = [torch.zeros(len(mine), dtype=torch.float64) for _ in range(self.world)]
        if self.collective:
            dist.all_gather(infos, mine, group=self.ctl)     # also orders "segment written" before "segment read"
        else:
            infos = [mine]
        self.last_busy_ms = [float(t[2]) for t in infos]
        failed = [r for r, t in enumerate(infos) if float(t[3]) > 0]
        if failed:                         # every rank leaves together; nobody keeps a lease on a result nobody will read
            if seg is not None:
                seg.lease = 0
            if err is not None:
                raise err
            raise RankFailure(f"rank {failed[0]} of the sharded conversion failed")
        if self.rank != 0:
            return None                    # the segment stays leased until rank 0 drops the assembled object
        mps = assemble(self.reader, [t.tolist() for t in infos], h["oc"], h["ucw"], {"busy_ms_per_rank": self.last_busy_ms})
        mps.info = dict(getattr(mps_l, "info", {}))
        allchk = np.array([t[4:].tolist() for t in infos])
        with np.errstate(all="ignore"):
            worst = np.where(np.isnan(allchk).all(axis=0), np.nan, np.nanmax(np.where(np.isnan(allchk), -np.inf, allchk), axis=0))
        mps.info["checks"] = {nm: float(v) for nm, v in zip(CHECK_NAMES, worst) if not np.isnan(v)}
        return mps

    def convert_local(self, C, trunc, ortho_center=None, unit_cell_width=None):
        """One conversion of this rank's site range, tensors landed.  Returns (generation, bytes, busy ms, checks)."""
        h = self.convert_begin(C, trunc, ortho_center, unit_cell_width, wait=True)
        if h["err"] is not None:
            raise h["err"]
        self.last_local = h["mps"]
        return h["seg"].gen, h["seg"].size, h["busy"], dict(getattr(h["mps"], "info", {}).get("checks", {}))

    def convert(self, C, trunc, ortho_center=None, unit_cell_width=None):
        """Launcher mode: the assembled ``MPSData`` on rank 0, ``None`` elsewhere."""
        return self.convert_end(self.convert_begin(C, trunc, ortho_center, unit_cell_width, wait=True))


def init_rank(local=None, gloo=False, dry=False):
    """Process-group set-up of one rank from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (launcher contract).
    local: device index (default LOCAL_RANK).  gloo: all ranks may share one device (RCCL refuses that): the
    one-GPU rehearsal, where C travels as a host tensor; the dry mode always runs on gloo.
    Returns (rank, world, device string or None)."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if local is None:
        local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dev = None
    if not dry:
        torch.cuda.set_device(local)
        dev = f"cuda:{local}"
    if not dist.is_initialized():
        dist.init_process_group("gloo" if (gloo or dry) else "nccl", rank=rank, world_size=world)
    return rank, world, dev


def make_engine(dev, dry=False):
    if dry:
        return DryEngine()
    from .engine import Engine
    return Engine(dev, profile=False)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(argv, world, extra_env=None, **popen_kw):
    """Starts ``world`` fresh Python processes with the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT) - what ``torch.distributed.run`` does, without its agent.  Must be called BEFORE
    the calling process initialises a GPU (a process that holds the device must not start others on this pool)."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TMF_SHM_TAG=f"tmf{os.getpid()}p{port}")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL between processes needs on this driver
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, **popen_kw))
    return procs


# ------------------------------------------------------------------------------------------------ library pool
class DevicePool:
    """Worker processes behind ``slater.C_to_MPS(..., devices=[...])``: one per device, started on first use -
    which has to come before this process initialises a GPU itself.  C goes to the workers through a shared-
    memory segment, their shards come back the same way; this process only maps and assembles."""

    def __init__(self, devices, timeout=600.0):
        import torch

        self.devices = [str(d) for d in devices]
        self.world = len(self.devices)
        self.dry = os.environ.get("TMF_DRY_ENGINE") == "1"
        if not self.dry and torch.cuda.is_initialized():
            raise RuntimeError("the device pool must be created before this process uses a GPU itself: call "
                               "slater.C_to_MPS(..., devices=[...]) (or multi_gpu.DevicePool) first")
        self.timeout = timeout
        self.sockpath = f"/tmp/tmf_pool_{os.getpid()}_{free_port()}.sock"
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(self.sockpath)
        srv.listen(self.world)
        srv.settimeout(timeout)
        ids = [d.split(":")[1] if ":" in d else "0" for d in self.devices]
        same = len(set(ids)) < len(ids)
        self.procs = spawn_ranks(["-m", "temfpy_amd.multi_gpu", "--serve", self.sockpath], self.world,
                                 extra_env={"TMF_POOL_DEVICES": ",".join(ids), "TMF_SAME_DEVICE": "1" if same else "0"})
        self.tag = None
        self.conns = [None] * self.world
        try:
            for _ in range(self.world):
                c, _a = srv.accept()
                c.settimeout(timeout)
                hello = _recv(c)
                self.conns[hello["rank"]] = c
                self.tag = hello["tag"]
        except Exception:
            self.close()
            raise
        finally:
            srv.close()
            try:
                os.unlink(self.sockpath)
            except OSError:
                pass
        self.reader = _Reader(self.tag)
        self.c_seg = None
        atexit.register(self.close)

    def convert(self, C, trunc, ortho_center=None, unit_cell_width=None):
        C = np.asarray(C)
        cplx = np.iscomplexobj(C)
        C = np.ascontiguousarray(C, np.complex128 if cplx else np.float64)
        L = len(C)
        if self.c_seg is None or self.c_seg.size < C.nbytes:
            if self.c_seg is not None:
                self.c_seg.unlink()
            self.c_gen = getattr(self, "c_gen", -1) + 1
            self.c_seg = Segment(f"{self.tag}_C{self.c_gen}", max(C.nbytes, 4096), create=True)
        self.c_seg.buf[: C.nbytes] = C.reshape(-1).view(np.uint8)
        try:
            blob = pickle.dumps(trunc)
        except Exception as exc:
            raise TypeError("the truncation parameters must be picklable to reach the worker processes "
                            f"(a lambda as `sectors`?): {exc}") from exc
        msg = dict(cmd="convert", c_name=self.c_seg.name, L=L, cplx=cplx, trunc=blob, oc=ortho_center, ucw=unit_cell_width)
        try:
            for c in self.conns:
                _send(c, msg)
            reps = self._collect()
        except (RuntimeError, TimeoutError, ConnectionError, OSError):
            self.discard()                 # a worker died or hangs: the ranks are no longer in step, the pool is unusable
            raise
        infos, checks, errs = [], {}, []
        for r, rep in enumerate(reps):
            if rep.get("error"):
                errs.append(rep)
            else:
                infos.append((r, rep["gen"], rep["size"], rep["busy_ms"]))
                checks.update(rep["checks"])
        if errs:
            # every worker has answered, so the group is in step and stays usable.  The ranks that succeeded hold a lease on
            # a result nobody will read: give it back.  Raise the real exception, not the RankFailure echo of the others.
            for r, gen, _size, _busy in infos:
                self.reader.segment(r, int(gen)).lease = 0
            real = [e for e in errs if e.get("type") != "RankFailure"] or errs
            raise _rebuild_exception(real[0])
        mps = assemble(self.reader, [i[1:] for i in infos], ortho_center or L // 2, unit_cell_width or L,
                       {"busy_ms_per_rank": [i[3] for i in infos]})
        mps.info = {"checks": checks}
        return mps

    def _collect(self):
        """One reply from every worker, whichever answers first; notices a dead worker or the deadline while waiting."""
        import select
        pending, replies = dict(enumerate(self.conns)), {}
        t_end = time.time() + self.timeout
        while pending:
            ready, _, _ = select.select(list(pending.values()), [], [], 0.5)
            for c in ready:
                r = next(k for k, v in pending.items() if v is c)
                replies[r] = _recv(c)
                del pending[r]
            if pending and not ready:
                dead = [r for r, p in enumerate(self.procs) if p.poll() is not None]
                if dead:
                    raise RuntimeError(f"worker of {self.devices[dead[0]]} exited with code {self.procs[dead[0]].returncode}")
                if time.time() > t_end:
                    raise TimeoutError(f"worker of {self.devices[min(pending)]} did not answer within {self.timeout} s")
        return [replies[r] for r in range(self.world)]

    def discard(self):
        """Ends a pool whose workers are out of step: processes killed, segments removed, no longer handed out by pool()."""
        for k in [k for k, v in _POOLS.items() if v is self]:
            del _POOLS[k]
        for c in getattr(self, "conns", []):
            if c is not None:
                try:
                    c.close()
                except OSError:
                    pass
        self.conns = []
        for p_ in getattr(self, "procs", []):
            if p_.poll() is None:
                p_.terminate()
        for p_ in getattr(self, "procs", []):
            try:
                p_.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p_.kill()
        self.procs = []
        if getattr(self, "c_seg", None) is not None:
            self.c_seg.unlink()
            self.c_seg = None
        if self.tag:                        # the workers' atexit handlers did not run
            for f in os.listdir("/dev/shm"):
                if f.startswith(self.tag + "_"):
                    try:
                        os.unlink("/dev/shm/" + f)
                    except OSError:
                        pass

    def close(self):
        for c in getattr(self, "conns", []):
            if c is not None:
                try:
                    _send(c, dict(cmd="exit"))
                    c.close()
                except OSError:
                    pass
        self.conns = []
        for p in getattr(self, "procs", []):
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        self.procs = []
        if getattr(self, "c_seg", None) is not None:
            self.c_seg.unlink()
            self.c_seg = None


def _send(conn, obj):
    b = pickle.dumps(obj)
    conn.sendall(len(b).to_bytes(8, "little") + b)


def _recv(conn):
    def exactly(n):
        out = b""
        while len(out) < n:
            chunk = conn.recv(n - len(out))
            if not chunk:
                raise ConnectionError("peer closed the pool connection")
            out += chunk
        return out
    n = int.from_bytes(exactly(8), "little")
    return pickle.loads(exactly(n))


def _rebuild_exception(rep):
    import builtins
    cls = (getattr(builtins, rep.get("type", ""), None) or getattr(np.linalg, rep.get("type", ""), None)
           or (RankFailure if rep.get("type") == "RankFailure" else None))
    if not (isinstance(cls, type) and issubclass(cls, Exception)):
        cls = RuntimeError
    return cls(f"{rep['error']} (rank {rep.get('rank')})")


def _serve(sockpath):
    """Worker main loop (``python -m temfpy_amd.multi_gpu --serve <socket>``)."""
    dry = os.environ.get("TMF_DRY_ENGINE") == "1"
    same = os.environ.get("TMF_SAME_DEVICE") == "1"
    ids = os.environ.get("TMF_POOL_DEVICES", "").split(",")
    rank, world, dev = init_rank(local=int(ids[int(os.environ["RANK"])] or 0), gloo=same, dry=dry)
    tag = os.environ["TMF_SHM_TAG"]
    group = ShardGroup(make_engine(dev, dry), tag, device=dev)
    conn = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    conn.connect(sockpath)
    _send(conn, dict(rank=rank, tag=tag))
    c_segs = {}
    while True:
        msg = _recv(conn)
        if msg["cmd"] == "exit":
            break
        try:
            C = None
            if rank == 0:
                seg = c_segs.get(msg["c_name"]) or c_segs.setdefault(msg["c_name"], Segment(msg["c_name"]))
                dt = np.complex128 if msg["cplx"] else np.float64
                C = np.frombuffer(seg.buf, dt, msg["L"] ** 2).reshape(msg["L"], msg["L"])
            gen, size, busy, checks = group.convert_local(C, pickle.loads(msg["trunc"]), msg["oc"], msg["ucw"])
            _send(conn, dict(gen=gen, size=size, busy_ms=busy, checks=checks))
        except Exception as exc:   # reported to the caller, which re-raises
            _send(conn, dict(error=str(exc), type=type(exc).__name__, rank=rank))
    import torch.distributed as dist
    dist.destroy_process_group()


_POOLS = {}


def pool(devices):
    key = tuple(str(d) for d in devices)
    if key not in _POOLS:
        _POOLS[key] = DevicePool(key)
    return _POOLS[key]


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--serve":
        _serve(sys.argv[2])
    else:
        raise SystemExit("usage: python -m temfpy_amd.multi_gpu --serve <socket>   (started by DevicePool)")
