import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import uniform_chain
from temfpy_amd import slater, gutzwiller
L, chi = int(sys.argv[1]), int(sys.argv[2])
C, N = slater.correlation_matrix(uniform_chain(L))
mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False, spinful="PH")
res = gutzwiller.abrikosov_ph(mps)
print("norm", res.norm, "chi", res.chi[::8])
for j, t in enumerate(res.dense_tensors()):
    X = np.einsum("pab,pcb->ac", t, t.conj())
    d = np.abs(X - np.eye(len(X)))
    if d.max() > 1e-10:
        i = np.unravel_index(d.argmax(), d.shape)
        print(f"site {j}: dev {d.max():.2e} at {i}, charges {res.charges[j][list(i)]}, lam {res.lam[j][list(i)]}, chi {len(X)}, diag dev {np.abs(np.diag(X)-1).max():.2e}")
