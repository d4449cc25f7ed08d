"""Read-only views of a conversion with the field names of the reference's intermediate dataclasses.

The reference exposes ``SchmidtModes`` (slater.py:42), ``SchmidtVectors`` (:495) and ``MPSTensorData`` (:873) as
public API (docs/source/reference/slater.rst:8-13).  Here their content is what the sweep keeps in HBM while it
runs; what reaches the host - and therefore these views - is everything except the dense orbital matrices and the
Schur complements: eigenvalues, orbital counts and column slices, occupation patterns, Schmidt values, charge
slices, merged-leg bookkeeping, ``det_always`` and the charge blocks.  ``vL`` / ``vR`` / ``sometimes_matrix`` are
``None`` (documented in DESIGN.md section 1): a caller that needs the orbitals themselves calls
``numpy.linalg.eigh`` on the block, as the reference does.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class SchmidtModes:
    """View of one cut with the fields of ``temfpy.slater.SchmidtModes`` (slater.py:42-134)."""
    e: np.ndarray
    ixL: dict
    ixR: dict
    nL: int
    nR: int
    n_fermion: int
    vL: None = None          # the orbital matrices stay on the device
    vR: None = None

    @property
    def n_entangled(self) -> int:
        return self.e.size

    def size(self, which: str = "T") -> int:
        return {"L": self.nL, "R": self.nR, "T": self.nL + self.nR}[which]

    def n_filled(self, which: str) -> int:
        ix = (self.ixL if which == "L" else self.ixR)["filled"]
        return ix.stop - ix.start

    def eigenvalues(self, which: str, entangled: bool = False) -> np.ndarray:
        """slater.py:210-250: all eigenvalues of C_LL / C_RR in the column order of vL / vR (filled = 1, empty = 0)."""
        eE = self.e if which == "L" else 1.0 - self.e[::-1]
        if entangled:
            return eE
        ix = self.ixL if which == "L" else self.ixR
        out = np.zeros(self.size(which))
        out[ix["filled"]] = 1.0
        out[ix["entangled"]] = eE
        return out

    @property
    def singular_values(self) -> np.ndarray:
        """Singular values of C_LR with the sign convention of slater.py:252-268."""
        k = self.n_entangled
        return np.sqrt(self.e * (1.0 - self.e)) * (-1.0) ** np.arange(k)[::-1]

    @property
    def e_ratio(self) -> np.ndarray:
        return np.log((1.0 - self.e) / self.e)

    def embed_subsets(self, sets: np.ndarray):
        """slater.py:430-470."""
        left = np.zeros((len(sets), self.nL), bool)
        left[:, self.ixL["entangled"]] = sets
        left[:, self.ixL["filled"]] = True
        right = np.zeros((len(sets), self.nR), bool)
        right[:, self.ixR["entangled"]] = np.logical_not(sets[:, ::-1])
        right[:, self.ixR["filled"]] = True
        return left, right

    def schmidt_values(self, sets: np.ndarray) -> np.ndarray:
        """slater.py:472-489."""
        return np.where(sets, self.e, 1.0 - self.e).prod(axis=1) ** 0.5

    @classmethod
    def from_bond(cls, bond, L: int):
        k, fl, fr = len(bond.e), bond.n_filled_left, bond.n_filled_right
        nL, nR = bond.x, L - bond.x
        # column order (slater.py:353-370): left [filled | entangled | empty], right [empty | entangled | filled]
        ixL = {"filled": slice(0, fl), "entangled": slice(fl, fl + k), "empty": slice(fl + k, nL)}
        ixR = {"empty": slice(0, nR - fr - k), "entangled": slice(nR - fr - k, nR - fr), "filled": slice(nR - fr, nR)}
        return cls(e=np.asarray(bond.e), ixL=ixL, ixR=ixR, nL=nL, nR=nR, n_fermion=fl + fr + k)


@dataclass(frozen=True)
class SchmidtVectors:
    """View of one cut with the fields of ``temfpy.slater.SchmidtVectors`` (slater.py:495-700)."""
    modes: SchmidtModes
    left_sets: np.ndarray
    right_sets: np.ndarray
    schmidt_values: np.ndarray       # unnormalised, as the reference stores them
    idx_L: dict

    n_schmidt = property(lambda self: len(self.schmidt_values))
    n_entangled = property(lambda self: self.modes.n_entangled)
    nL = property(lambda self: self.modes.nL)
    nR = property(lambda self: self.modes.nR)
    n_fermion = property(lambda self: self.modes.n_fermion)
    vL = property(lambda self: None)
    vR = property(lambda self: None)

    def size(self, which: str = "T") -> int:
        return self.modes.size(which)

    def sets(self, which: str) -> np.ndarray:
        return self.left_sets if which == "L" else self.right_sets

    @classmethod
    def from_bond(cls, bond, L: int):
        modes = SchmidtModes.from_bond(bond, L)
        left, right = modes.embed_subsets(bond.sets)
        return cls(modes=modes, left_sets=left, right_sets=right, schmidt_values=np.asarray(bond.lam_raw), idx_L=bond.idx_L)


@dataclass(frozen=True)
class MPSTensorData:
    """View of one site with the fields of ``temfpy.slater.MPSTensorData`` (slater.py:873-1104) that reach the host.

    ``blocks`` maps the ket charge to the dense sector ``det_always * _tensor_block(...)`` - the arrays the reference
    assigns at slater.py:1137-1141 - with rows on the merged (p, bra) leg in the order of ``bra_p`` / ``bra_alpha``."""
    mode: str
    physical_leg: bool
    det_always: complex
    idx_bra: dict            # charge slices of the bra leg BEFORE the physical leg is merged in (slater.py:1099)
    idx_ket: dict
    merged_idx: dict         # charge slices of the merged (p, bra) leg: what LegPipe.to_qdict() gives at slater.py:1131
    qtotal: int
    bra_p: np.ndarray
    bra_alpha: np.ndarray
    blocks: dict
    sometimes_matrix: None = None    # the Schur complement stays on the device

    @classmethod
    def from_site(cls, mps, i: int):
        s = mps.sites[i]
        bra, ket = (mps.bonds[i], mps.bonds[i + 1]) if s.mode == "left" else (mps.bonds[i + 1], mps.bonds[i])
        q_rows = np.asarray(bra.q_left)[s.bra_alpha] + (s.bra_p if s.mode == "left" else -s.bra_p)
        # merged leg: ascending left charge in both modes (slater.py:1053-1058)
        qs, start = np.unique(q_rows, return_index=True)
        order = np.argsort(start)
        qs, start = qs[order], start[order]
        stop = np.concatenate((start[1:], [len(q_rows)]))
        merged = {int(q): slice(int(a), int(b)) for q, a, b in zip(qs, start, stop)}
        return cls(mode=s.mode, physical_leg=True, det_always=s.det_always, idx_bra=bra.idx_L, idx_ket=ket.idx_L,
                   merged_idx=merged, qtotal=s.qtotal, bra_p=s.bra_p, bra_alpha=s.bra_alpha,
                   blocks={int(b[0]): b[5] for b in s.blocks})
