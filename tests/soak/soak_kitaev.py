"""Randomised soak of the Pfaffian path on Kitaev-like chains (dimerised Majorana couplings, both phases, optional disorder):
cuts through the topological phase carry eigenvalue-1/2 modes, where only gauge-invariant quantities are compared
(check_against_oracle(..., degenerate=True) of tests/test_gpu_pfaffian.py).  Development aid.
usage: python tests/soak/soak_kitaev.py [cases] [first seed]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_pfaffian as tp  # noqa: E402
from temfpy_amd import pfaffian  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = refused = 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(3, 17))
    t1, t2 = rng.uniform(0.3, 2.0), rng.uniform(0.3, 2.0)
    dis = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    M = np.zeros(2 * L - 1)
    M[0::2], M[1::2] = t1, t2
    M = M * (1 + dis * rng.uniform(-1, 1, 2 * L - 1))
    H = 1j * (np.diag(M, 1) - np.diag(M, -1))
    chi = int(rng.choice([4, 16, 64]))
    oc = int(rng.integers(1, L)) if rng.integers(0, 2) else None
    tag = f"seed {seed}: L={L} t=({t1:.2f},{t2:.2f}) disorder={dis} chi={chi} oc={oc}"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                C = pfaffian.correlation_matrix(H, basis="M->M")
            except RuntimeError:        # zero modes of the whole chain: the reference refuses as well (pfaffian.py:372-377)
                refused += 1
                continue
            mps = pfaffian.C_to_MPS(C, {"chi_max": chi}, basis="M", ortho_center=oc)
            try:
                cuts, _sites = tp.porc.c_to_mps(C, {"chi_max": chi}, oc)
            except AssertionError:      # the reference's own consistency assertions (pfaffian.py:795-805) fail on an eigenvalue
                refused += 1            # within rounding of the cutoff: it refuses the input, nothing to compare with
                continue
            event = False
            for b in range(L + 1):
                ea, eo = np.asarray(mps.bonds[b].e), np.asarray(cuts[b].e)
                if ea.shape != eo.shape:
                    # threshold event (i) of DESIGN section 2: an eigenvalue within rounding of the cutoff svd_min^2 = 1e-12
                    big = eo if len(eo) > len(ea) else ea
                    if np.abs(big - 1e-12).min() < 1e-15:
                        event = True
                        continue
                if ea.shape != eo.shape or np.abs(ea - eo).max(initial=0) > 1e-12:
                    raise AssertionError(f"bond {b}: eigenvalues differ ({ea.shape} vs {eo.shape})")
                if event:
                    continue
                if (mps.bonds[b].pL + mps.bonds[b].pR) % 2 != (cuts[b].pL + cuts[b].pR) % 2:
                    raise AssertionError(f"bond {b}: total parity differs")
                la, lo = np.sort(mps.bonds[b].lam), np.sort(cuts[b].lam)
                # (1e-9 as on the Slater path: a Schmidt value of 1e-6 from an orbital next to the 1e-12 cutoff carries the
                # 1e-4 relative noise of that eigenvalue)
                if la.shape != lo.shape or np.abs(la - lo).max() > 1e-9:
                    raise AssertionError(f"bond {b}: Schmidt values differ by {np.abs(la - lo).max() if la.shape == lo.shape else 'count'}")
    except Exception as e:          # noqa: BLE001
        bad += 1
        msg = " ".join(x.strip() for x in str(e).splitlines() if "Max absolute" in x or "Mismatched" in x or "assert" in x.lower())
        print("MISMATCH", tag, "->", type(e).__name__, (msg or str(e))[:160], flush=True)
print(f"{n_cases} cases, {bad} mismatches, {refused} chains with zero modes refused")
sys.exit(1 if bad else 0)
