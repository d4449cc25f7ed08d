import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import slater_oracle as orc
from tests_inputs import uniform_chain
from temfpy_amd import slater
L = int(sys.argv[1]) if len(sys.argv) > 1 else 48
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 96
C, _ = orc.correlation_matrix(uniform_chain(L))
mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False, spinful="PH")
cuts, sites = orc.c_to_mps(C, {"chi_max": chi}, spinful="PH")
Sr, Sh = orc.entropies(cuts), mps.entanglement_entropy(all_bonds=True)
for b in range(2 * L + 1):
    c, m = cuts[b], mps.bonds[b]
    flag = ""
    if c.k != len(m.e) or c.n_filled("L") != m.n_filled_left or c.n_filled("R") != m.n_filled_right:
        flag = "  <-- k/nf mismatch"
    elif np.abs(c.e - m.e).max(initial=0) > 1e-12:
        flag = f"  <-- e mismatch {np.abs(c.e - m.e).max():.2e}"
    if flag or abs(Sr[b] - Sh[b]) > 1e-3:
        print(b, "k", c.k, len(m.e), "nfL", c.n_filled("L"), m.n_filled_left, "nfR", c.n_filled("R"), m.n_filled_right,
              "chi", len(c.lam), m.chi, f"dS={Sr[b]-Sh[b]:+.3e}", flag)
        if flag:
            print("    e ref", c.e[:8], "\n    e hip", m.e[:8])
