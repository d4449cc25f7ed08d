"""Deviations that the reference's own check_schmidt_decomposition (testing.py:131-177) evaluates at the
centre cut, from the reference's SchmidtModes (imported as in make_golden.py).  Output: ref_checks.json
(inputs are regenerated in the test from tests_inputs.random_hopping).  Run in the build container only."""
import json, os, sys, warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE), sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
from tests_inputs import random_hopping  # noqa: E402

slater, testing = mg.load_reference()
testing.TEST_ACTION = "warn"
out = {}
for L, seed, chi in [(32, 3, 64), (64, 1, 64), (96, 0, 128)]:
    C, _ = slater.correlation_matrix(random_hopping(L, seed))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m = slater.SchmidtModes.from_correlation_matrix(C, L // 2, {"chi_max": chi}, which="LR")
    warned = sorted({str(w.message).strip().split("\n")[0] for w in rec if type(w.message).__name__ == "ComparisonWarning"})
    N = L // 2
    HT = lambda a: a.conj().T  # noqa: E731
    dev = {
        "vL is not unitary": float(np.abs(m.vL @ HT(m.vL) - np.eye(N)).max()),
        "vL does not diagonalise C_LL": float(np.abs((m.eigenvalues("L") * m.vL) @ HT(m.vL) - C[:N, :N]).max()),
        "vR is not unitary": float(np.abs(m.vR @ HT(m.vR) - np.eye(L - N)).max()),
        "vR does not diagonalise C_RR": float(np.abs((m.eigenvalues("R") * m.vR) @ HT(m.vR) - C[N:, N:]).max()),
        "vL and vR do not SVD C_LR": float(np.abs((m.singular_values * m.vL_entangled) @ HT(m.vR_entangled[:, ::-1]) - C[:N, N:]).max()),
    }
    out[f"rand_L{L}_s{seed}_chi{chi}"] = {"L": L, "seed": seed, "chi": chi, "deviations": dev, "warned": warned}
    print(L, seed, dev, warned)
json.dump(out, open(os.path.join(HERE, "ref_checks.json"), "w"), indent=1)
