"""Randomised soak of slater.C_to_iMPS (determinant construction, slater.py:1499-1563) against the oracle's restatement and
against the package's own finite conversions: gapped chains with a random unit cell (2 - 4 sites, random hoppings and on-site
energies, optional Peierls phase), random length, cut, chi and spin mode.  Checks per case: (i) Schmidt values of every bond of
the cell against the oracle (1e-9), (ii) the same infinite state as the oracle's cell - dominant eigenvalue of the mixed
transfer matrix 1 to 1e-7 (spinless cases), (iii) the acceptance check of src/examples/iMPS.py:27-38 with a SEPARATELY converted
short chain: two sweeps number and sign the Schmidt vectors of the cut alike (1e-6).  Development aid.
usage: python tests/soak/soak_c_to_imps.py [cases] [first seed]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import imps_oracle as io  # noqa: E402
from oracle import slater_oracle as orc  # noqa: E402
from temfpy_amd import slater  # noqa: E402


def dense(m):
    return m.dense_tensors(), [np.asarray(x) for x in m.lam], list(m.form)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = skipped = 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    cell = int(rng.choice([2, 2, 3, 4]))
    hop = -rng.uniform(0.4, 2.0, cell)
    hop[int(rng.integers(0, cell))] *= 2.5           # one strong bond per cell: gapped at the fillings used
    mu = rng.uniform(-0.3, 0.3, cell) * float(rng.choice([0.0, 1.0]))
    imag = float(rng.choice([0.0, 0.0, 0.4]))
    ncell = int(rng.integers(5, 13))
    L = cell * ncell
    cut = cell * int(rng.integers(2, ncell - 1))
    chi = int(rng.choice([16, 40, 96]))
    spinful = [None, None, None, "simple", "PH"][int(rng.integers(0, 5))]
    tag = f"seed {seed}: cell={cell} hop={np.round(hop, 2)} mu={np.round(mu, 2)} phase={imag} L={L} cut={cut} chi={chi} spinful={spinful}"

    def ham(n):
        M = np.array([hop[i % cell] for i in range(n - 1)], complex if imag else float)
        if imag:
            M = M * np.exp(1j * imag)
        H = np.diag(M, 1)
        H = H + H.conj().T + np.diag([mu[i % cell] for i in range(n)])
        return H
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            N = (L // cell) * (cell // 2 if cell % 2 == 0 else 1)      # the same filling per cell in all chains
            per_cell = N // (L // cell)
            try:
                Cs, _ = slater.correlation_matrix(ham(L), per_cell * (L // cell))
                Cl, _ = slater.correlation_matrix(ham(L + cell), per_cell * (L // cell + 1))
                Cv, _ = slater.correlation_matrix(ham(L + cell * 2), per_cell * (L // cell + 2))
            except ValueError:      # no gap at this filling (edge states at the Fermi level): not a unique Slater determinant
                skipped += 1
                continue
            res, err = slater.C_to_iMPS(Cs, Cl, {"chi_max": chi}, cell, cut, spinful=spinful, as_tenpy=False)
            mult = 1 if spinful is None else 2
            if spinful is None:
                To, So, (lu, ls_), G = orc.c_to_imps(Cs, Cl, {"chi_max": chi}, cell, cut)
                for b, (a, o) in enumerate(zip(res.lam, So)):
                    if len(a) != len(o) or np.abs(np.sort(a) - np.sort(o)).max() > 1e-9:
                        raise AssertionError(f"bond {b} of the cell: Schmidt values differ")
                E = None
                for a_, b_ in zip(res.dense_tensors(), To):
                    step = np.einsum("pab,pcd->acbd", a_.conj(), b_).reshape(a_.shape[1] * b_.shape[1], a_.shape[2] * b_.shape[2])
                    E = step if E is None else E @ step
                eta = np.abs(np.linalg.eigvals(E)).max()
                if abs(eta - 1) > 1e-7:
                    raise AssertionError(f"mixed transfer matrix with the oracle's cell: |eta| = {eta}")
            # acceptance with separately converted chains
            n_ins = 2
            ms = slater.C_to_MPS(Cs, {"chi_max": chi}, ortho_center=mult * cut, spinful=spinful, as_tenpy=False)
            mv = slater.C_to_MPS(Cv, {"chi_max": chi}, ortho_center=mult * cut, spinful=spinful, as_tenpy=False)
            Ts, ls, fs = dense(ms)
            Tr, lr, fr = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, mult * cut, n_ins)
            Tv, lv, fv = dense(mv)
            ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
            nr, nv = io.overlap(Tr, lr, fr, Tr, lr, fr).real, io.overlap(Tv, lv, fv, Tv, lv, fv).real
            dev = abs(abs(ov) / np.sqrt(nr * nv) - 1)
            # (the truncated chains differ from each other by what chi_max cuts away: the bound follows the discarded weight)
            if dev > max(1e-6, 50 * err.left_unitary ** 2):
                raise AssertionError(f"reconstruction overlap 1 - {dev:.2e} (left errors {err.left_unitary:.1e}, {err.left_schmidt:.1e})")
    except Exception as e:          # noqa: BLE001
        bad += 1
        print("MISMATCH", tag, "->", type(e).__name__, str(e)[:200], flush=True)
print(f"{n_cases} cases, {bad} mismatches, {skipped} without a gap at the filling")
sys.exit(1 if bad else 0)
