"""Backend-neutral result of the sweep: what ``tenpy.networks.mps.MPS`` would hold.

``slater.C_to_MPS`` returns a TeNPy ``MPS`` when TeNPy is importable (``to_tenpy``), else
this object: per-site charge blocks of the rank-3 tensors, per-bond Schmidt values and
charges, and the canonical form list (slater.py:1348-1351).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class BondData:
    """Schmidt data of one entanglement cut (``SchmidtVectors``, slater.py:494-700)."""
    x: int
    e: np.ndarray            # entangled eigenvalues of C_LL, descending (SchmidtModes.e)
    n_filled_left: int
    n_filled_right: int
    masks: np.ndarray        # (chi, 2) uint64: bit i <-> entangled orbital i occupied on the left
    lam_raw: np.ndarray      # unnormalised Schmidt values (SchmidtVectors.schmidt_values)
    lam: np.ndarray          # normalised (utils.normalize_SV)
    q_left: np.ndarray       # particles to the left of the cut, ascending
    n_checked: int = 0

    @property
    def chi(self):
        return len(self.lam)

    @property
    def sets(self):
        """(chi, k) bool array as ``SchmidtVectors`` stores it (slater.py:438-443)."""
        k = len(self.e)
        out = np.zeros((len(self.masks), k), bool)
        for i in range(k):
            out[:, i] = (self.masks[:, i // 64] >> np.uint64(i % 64)) & np.uint64(1)
        return out

    @property
    def idx_L(self):
        """{N_left: slice} as ``SchmidtVectors.idx_L`` (slater.py:681-683)."""
        q, start = np.unique(self.q_left, return_index=True)
        stop = np.concatenate((start[1:], [len(self.q_left)]))
        return {int(a): slice(int(b), int(c)) for a, b, c in zip(q, start, stop)}


@dataclass
class SiteData:
    """One MPS tensor as dense charge blocks (``MPSTensorData.to_npc_array``, slater.py:1106-1143)."""
    mode: str                # "left": A[p, vL, vR]; "right": B[p, vL, vR]
    det_always: complex
    qtotal: int
    bra_p: np.ndarray        # physical occupation of each row of the merged (p, bra) leg
    bra_alpha: np.ndarray    # bra Schmidt-vector index of each merged row
    blocks: list = field(default_factory=list)  # (q_ket, r0, r1, c0, c1, ndarray (r1-r0, c1-c0))
    chi_bra: int = 0
    chi_ket: int = 0

    def dense(self):
        """(2, chi_left, chi_right) dense tensor."""
        dt = self.blocks[0][5].dtype if self.blocks else float
        T = np.zeros((2, self.chi_bra, self.chi_ket), dt)
        for _, r0, r1, c0, c1, blk in self.blocks:
            T[self.bra_p[r0:r1], self.bra_alpha[r0:r1], c0:c1] = blk
        return T if self.mode == "left" else T.transpose(0, 2, 1)

    def norm(self):
        return float(np.sqrt(sum((np.abs(b[5]) ** 2).sum() for b in self.blocks)))


class LazyList:
    """List-like view of the bonds (or sites) of a conversion: the objects are built on first access from
    the flat arrays the sweep produced (building all 1025 objects eagerly cost 4 ms of Python per
    conversion, on the critical path once the determinant stage had shrunk to 3 ms)."""

    def __init__(self, n, factory):
        self._n, self._factory, self._cache = n, factory, {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if i not in self._cache:
            self._cache[i] = self._factory(i)
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(self._n))


_ALIGN = 4096


class ShardArrays:
    """Flat result of one conversion, or of one rank's site range of it: what the sweep leaves in host
    memory.  ``arrays`` maps names to NumPy arrays, ``meta`` holds scalars.  The tensors are ONE flat
    array (``out``: the charge blocks of all sites back to back, row-major per block), which is what the
    GPU writes through page-locked memory; everything else is integer bookkeeping of the host phase.

    bonds   my_cuts (ncut), c_sets (ncut, cap, 2) u64, c_lam / c_q (ncut, cap), c_chi, c_chk,
            e_pool + e_off + kk_cut (entangled eigenvalues), nfl / nfr (filled orbitals left / right)
    sites   mode, sec_off, nsec, sectors (tmf_sector records), out_off, bra_off, chi_b, chi_k, bra_p,
            bra_alpha, det (det_always per site), out
    meta    L, s_lo, s_hi, ortho_center

    ``pack`` / ``unpack`` move the whole object through one byte buffer (a shared-memory segment of the
    multi-GPU path): [u64 header length][JSON header][arrays at 4096-byte aligned offsets]."""

    def __init__(self, arrays, meta, keepalive=None):
        self.arrays, self.meta, self.keepalive = arrays, meta, keepalive
        self._cpos = None
        self.wait = None      # callable that blocks until ``out`` / ``det`` have landed (asynchronous download)

    # ---- serialisation ---------------------------------------------------------------------------
    HEADER_ROOM = 16384

    @staticmethod
    def plan(spec):
        """spec: {name: (dtype, shape)} -> ({name: [dtype descr, shape, byte offset]}, total bytes)."""
        entries, off = {}, ShardArrays.HEADER_ROOM
        for name, (dt, shape) in spec.items():
            dt = np.dtype(dt)
            off = (off + _ALIGN - 1) & ~(_ALIGN - 1)
            entries[name] = [dt.descr if dt.fields else dt.str, [int(x) for x in shape], off]
            off += int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        return entries, (off + _ALIGN - 1) & ~(_ALIGN - 1)

    @staticmethod
    def _views(buf, entries):
        arrays = {}
        for name, (dt, shape, off) in entries.items():
            dt = np.dtype([tuple(f) for f in dt]) if isinstance(dt, list) else np.dtype(dt)
            arrays[name] = np.frombuffer(buf, dt, int(np.prod(shape, dtype=np.int64)), off).reshape(shape)
        return arrays

    @classmethod
    def create(cls, buf, entries, meta, keepalive=None):
        """Header into ``buf`` (uint8 array of at least ``plan``'s size); the arrays are uninitialised views."""
        import json
        hdr = json.dumps({"meta": meta, "arrays": entries}).encode()
        assert len(hdr) + 8 <= cls.HEADER_ROOM - 16, "header of the packed shard result too long"
        buf[:8] = np.frombuffer(np.uint64(len(hdr)).tobytes(), np.uint8)
        buf[8: 8 + len(hdr)] = np.frombuffer(hdr, np.uint8)
        return cls(cls._views(buf, entries), dict(meta), keepalive)

    @classmethod
    def unpack(cls, buf, keepalive=None):
        """Zero-copy views into a buffer written by ``create`` (kept alive through ``keepalive``)."""
        import json
        n = int(np.frombuffer(bytes(buf[:8]), np.uint64)[0])
        hdr = json.loads(bytes(buf[8: 8 + n]).decode())
        return cls(cls._views(buf, hdr["arrays"]), hdr["meta"], keepalive)

    # ---- object construction ---------------------------------------------------------------------
    @property
    def cpos(self):
        if self._cpos is None:
            c = np.full(int(self.meta["L"]) + 1, -1, np.int64)
            c[self.arrays["my_cuts"]] = np.arange(len(self.arrays["my_cuts"]))
            self._cpos = c
        return self._cpos

    def has_bond(self, b):
        return self.cpos[b] >= 0

    def has_site(self, i):
        return self.meta["s_lo"] <= i < self.meta["s_hi"]

    def bond(self, b):
        A = self.arrays
        j = int(self.cpos[b])
        if j < 0:
            return None
        ch = int(A["c_chi"][j])
        lam_raw = A["c_lam"][j, :ch]
        o, kk = int(A["e_off"][j]), int(A["kk_cut"][j])
        return BondData(x=b, e=A["e_pool"][o: o + kk], n_filled_left=int(A["nfl"][j]), n_filled_right=int(A["nfr"][j]),
                        masks=A["c_sets"][j, :ch], lam_raw=lam_raw, lam=lam_raw / np.sqrt(np.dot(lam_raw, lam_raw)),
                        q_left=A["c_q"][j, :ch], n_checked=int(A["c_chk"][j]))

    def site(self, i):
        if not self.has_site(i):
            return None
        if self.wait is not None:
            self.wait()
        A = self.arrays
        j = i - int(self.meta["s_lo"])
        h_out = A["out"]
        so_, oo = int(A["sec_off"][j]), int(A["out_off"][j])
        blocks = []
        for sec in A["sectors"][so_: so_ + int(A["nsec"][j])]:
            r0, r1, c0, c1 = (int(sec[f]) for f in ("r0", "r1", "c0", "c1"))
            o = oo + int(sec["out_off"])
            blocks.append((int(sec["q"]), r0, r1, c0, c1, h_out[o: o + (r1 - r0) * (c1 - c0)].reshape(r1 - r0, c1 - c0)))
        bo, cb = int(A["bra_off"][j]), int(A["chi_b"][j])
        return SiteData(mode="left" if A["mode"][j] == 0 else "right", det_always=A["det"][j], qtotal=0,
                        bra_p=A["bra_p"][bo: bo + 2 * cb], bra_alpha=A["bra_alpha"][bo: bo + 2 * cb], blocks=blocks,
                        chi_bra=cb, chi_ket=int(A["chi_k"][j]))

    def check_det(self):
        """The overlap of the always-occupied orbitals of two neighbouring cuts is singular: the reference
        fails in numpy.linalg.inv at slater.py:1079 / :1086 with the same exception."""
        d = self.arrays["det"]
        if not np.all(np.isfinite(d)) or np.any(d == 0):
            raise np.linalg.LinAlgError("Singular matrix")


class MPSData:
    """Finite MPS in mixed canonical form A..A [lam] B..B with U(1) charge blocks."""

    def __init__(self, bonds, sites, ortho_center, unit_cell_width, timings=None):
        self.bonds = bonds
        self.sites = sites
        self.L = len(bonds) - 1
        self.ortho_center = ortho_center
        self.unit_cell_width = unit_cell_width
        self.form = ["A"] * ortho_center + ["B"] * (self.L - ortho_center)  # slater.py:1348
        self.timings = timings or {}
        self.shards = []

    @classmethod
    def from_shards(cls, shards, ortho_center, unit_cell_width, timings=None, with_sites=True):
        """One MPS from the flat results of one or several site ranges (ranks): a bond / site object is
        built on first access from the arrays of the shard that owns it (a cut on a shard boundary is
        held, bit-identically, by both neighbours; the lower rank's copy is used)."""
        L = int(shards[0].meta["L"])

        def bond(b):
            for s in shards:
                if s.has_bond(b):
                    return s.bond(b)
            return None

        def site(i):
            for s in shards:
                if s.has_site(i):
                    return s.site(i)
            return None

        res = cls(LazyList(L + 1, bond), LazyList(L, site) if with_sites else [], ortho_center, unit_cell_width, timings)
        res.shards = list(shards)
        return res

    def schmidt_modes(self, bond: int):
        """``temfpy.slater.SchmidtModes`` view of one cut (slater.py:42; orbital matrices not downloaded)."""
        from .views import SchmidtModes
        return SchmidtModes.from_bond(self.bonds[bond], self.L)

    def schmidt_vectors(self, bond: int):
        """``temfpy.slater.SchmidtVectors`` view of one cut (slater.py:495)."""
        from .views import SchmidtVectors
        return SchmidtVectors.from_bond(self.bonds[bond], self.L)

    def tensor_data(self, site: int):
        """``temfpy.slater.MPSTensorData`` view of one site (slater.py:873; Schur complement not downloaded)."""
        from .views import MPSTensorData
        return MPSTensorData.from_site(self, site)

    def wait(self):
        """Blocks until the tensors of an asynchronous download have landed in host memory."""
        for s in self.shards:
            if s.wait is not None:
                s.wait()
        return self

    @property
    def lam(self):
        return [b.lam for b in self.bonds]

    @property
    def chi(self):
        return [b.chi for b in self.bonds]

    def entanglement_entropy(self, all_bonds=False):
        """S(b) = -sum lam^2 ln lam^2; TeNPy's default omits the two trivial outer bonds."""
        out = np.zeros(self.L + 1)
        for i, b in enumerate(self.bonds):
            p = b.lam**2
            p = p[p > 0]
            out[i] = -(p * np.log(p)).sum()
        return out if all_bonds else out[1:-1]

    def dense_tensors(self):
        return [s.dense() for s in self.sites]

    def to_tenpy(self, verify=True):
        """Assemble ``tenpy.networks.mps.MPS`` exactly like slater.py:1106-1143,1348-1351.

        Needs physics-tenpy, which is not installed in the build environment: the row order of TeNPy's
        ``LegPipe`` (slater.py:1118) could not be pinned there.  The assembly therefore CHECKS ITSELF at run time
        (``verify``): every assembled tensor (a sample of 16 sites on long chains) is read back through
        ``to_ndarray()`` and compared element by element with this object's own dense tensor; a mismatch raises
        instead of handing a permuted tensor to TeNPy.  ``slater.C_to_MPS(..., as_tenpy=False)`` skips TeNPy."""
        import tenpy.linalg.np_conserved as npc  # noqa: F401  (ImportError is the caller's signal)
        from tenpy import networks

        self.wait()
        site = networks.site.FermionSite()
        leg_p = site.leg
        chinfo = leg_p.chinfo
        dtype = next((s.blocks[0][5].dtype for s in self.sites if s.blocks), np.dtype(float))
        check = set(range(self.L)) if self.L <= 64 else set(np.linspace(0, self.L - 1, 16).astype(int).tolist())
        tensors = []
        for i, s in enumerate(self.sites):
            left = s.mode == "left"
            bra, ket = (self.bonds[i], self.bonds[i + 1]) if left else (self.bonds[i + 1], self.bonds[i])
            qconj = (+1, -1) if left else (-1, +1)
            names = ("vL", "vR") if left else ("vR", "vL")
            leg_bra = npc.LegCharge.from_qdict(chinfo, bra.idx_L, qconj=qconj[0])
            leg_ket = npc.LegCharge.from_qdict(chinfo, ket.idx_L, qconj=qconj[1])
            pipe = npc.LegPipe([leg_p, leg_bra], qconj=leg_bra.qconj)
            B = npc.zeros([pipe, leg_ket], labels=[f"(p.{names[0]})", names[1]], dtype=dtype, qtotal=(s.qtotal,))
            qd = pipe.to_qdict()
            for q, r0, r1, c0, c1, blk in s.blocks:
                B[qd[(q + s.qtotal * qconj[0],)], slice(c0, c1)] = blk
            B = B.split_legs()
            if verify and i in check:
                got = B.to_ndarray()                                   # (p, bra, ket)
                want = s.dense() if left else s.dense().transpose(0, 2, 1)
                if got.shape != want.shape or not np.array_equal(got, want):
                    raise RuntimeError(
                        f"TeNPy assembly self-check failed at site {i}: the tensor read back from np_conserved differs "
                        "from the converter's own (LegPipe row order other than slater.py:943-952 documents?). "
                        "Use slater.C_to_MPS(..., as_tenpy=False) for the backend-neutral MPSData.")
            tensors.append(B)
        psi = networks.mps.MPS([site] * self.L, tensors, self.lam, form=self.form,
                               unit_cell_width=self.unit_cell_width)
        psi._temfpy_amd = self      # gutzwiller / iMPS entry points of this package continue from the device-side layout
        return psi
