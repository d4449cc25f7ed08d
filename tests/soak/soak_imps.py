"""Randomised soak of iMPS.MPS_to_iMPS against the CPU oracle fed with the same two finite MPS: dimerised chains with random
hoppings (gapped), random length, cut, chi, Peierls phase and orthogonality centres.  Development aid.
usage: python tests/soak/soak_imps.py [cases] [first seed]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import imps_oracle as io  # noqa: E402
from temfpy_amd import iMPS, slater  # noqa: E402


def dense(m):
    return m.dense_tensors(), [np.asarray(x) for x in m.lam], list(m.form)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    t1, t2 = -rng.uniform(1.2, 2.0), -rng.uniform(0.5, 1.0)
    if rng.integers(0, 2):
        t1, t2 = t2, t1
    imag = float(rng.choice([0.0, 0.0, 0.3, 1.1]))
    L = 2 * int(rng.integers(8, 21))
    cut = 2 * int(rng.integers(3, L // 2 - 2))
    chi = int(rng.choice([12, 32, 64]))
    ocs = None if rng.integers(0, 2) else int(rng.integers(1, L))
    ocl = None if rng.integers(0, 2) else int(rng.integers(1, L + 2))
    tag = f"seed {seed}: t=({t1:.2f},{t2:.2f}) phase={imag} L={L} cut={cut} chi={chi} oc=({ocs},{ocl})"

    def ham(n):
        M = t1 * np.ones(n - 1, complex if imag else float)
        M[1::2] = t2
        if imag:
            M = M * np.exp(1j * imag)
        M = np.diag(M, 1)
        return M + M.conj().T

    def finite(n, oc):
        C, _ = slater.correlation_matrix(ham(n))
        return slater.C_to_MPS(C, {"chi_max": chi}, ortho_center=oc, as_tenpy=False)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ms, ml = finite(L, ocs), finite(L + 2, ocl)
            res, err = iMPS.MPS_to_iMPS(ms, ml, 2, cut, offset=0)
            B, S, eo = io.mps_to_imps(*dense(ms), *dense(ml), 2, cut)
        # The Procrustes rotations in directions whose Schmidt weight is at the truncation threshold are fixed by rounding only
        # (LAPACK completes them to some unitary, the Jacobi SVD of the weighted overlap drops singular values below 1e-14
        # of the largest): the Schmidt-mixing measures and the weighted tensors see those directions at the size of the weight.
        weak = 30 * float(min(np.min(ms.lam[cut]), np.min(ml.lam[cut]), np.min(ml.lam[cut + 2])))
        dev_ = np.abs(np.asarray(list(err)) - np.asarray(eo))
        if max(dev_[0], dev_[2]) > 1e-8 or max(dev_[1], dev_[3]) > 1e-8 + weak:
            raise AssertionError(f"error metrics {list(err)} vs {list(eo)} (weak directions {weak:.1e})")
        for a, b in zip(res.lam, S):
            if len(a) != len(b) or np.abs(np.asarray(a) - np.asarray(b)).max() > 1e-12:
                raise AssertionError("Schmidt values of the cell differ")
        for t, r, sl in zip(res.dense_tensors(), B, S[:-1]):
            w = (sl[None, :, None] * np.abs(t - r)).max()
            if w > 1e-8 + weak:
                raise AssertionError(f"weighted tensor deviation {w:.1e}")
    except Exception as e:          # noqa: BLE001
        bad += 1
        print("MISMATCH", tag, "->", type(e).__name__, str(e)[:200], flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
