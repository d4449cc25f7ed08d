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

    def to_tenpy(self):
        """Assemble ``tenpy.networks.mps.MPS`` exactly like slater.py:1106-1143,1348-1351.

        Needs physics-tenpy; it is not installed in the build environment, so this
        assembly is untested there ('parity unpinned: LegPipe order')."""
        import tenpy.linalg.np_conserved as npc  # noqa: F401  (ImportError is the caller's signal)
        from tenpy import networks

        site = networks.site.FermionSite()
        leg_p = site.leg
        chinfo = leg_p.chinfo
        tensors = []
        for i, s in enumerate(self.sites):
            left = s.mode == "left"
            bra, ket = (self.bonds[i], self.bonds[i + 1]) if left else (self.bonds[i + 1], self.bonds[i])
            qconj = (+1, -1) if left else (-1, +1)
            names = ("vL", "vR") if left else ("vR", "vL")
            leg_bra = npc.LegCharge.from_qdict(chinfo, bra.idx_L, qconj=qconj[0])
            leg_ket = npc.LegCharge.from_qdict(chinfo, ket.idx_L, qconj=qconj[1])
            pipe = npc.LegPipe([leg_p, leg_bra], qconj=leg_bra.qconj)
            B = npc.zeros([pipe, leg_ket], labels=[f"(p.{names[0]})", names[1]], dtype=s.blocks[0][5].dtype,
                          qtotal=(s.qtotal,))
            qd = pipe.to_qdict()
            for q, r0, r1, c0, c1, blk in s.blocks:
                B[qd[(q + s.qtotal * qconj[0],)], slice(c0, c1)] = blk
            tensors.append(B.split_legs())
        return networks.mps.MPS([site] * self.L, tensors, self.lam, form=self.form,
                                unit_cell_width=self.unit_cell_width)
