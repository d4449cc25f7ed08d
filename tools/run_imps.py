"""slater.C_to_iMPS on a longer dimerised chain: the determinant construction (default) against the round-2 method
(TMF_IMPS=transfer), wall time and the acceptance check of src/examples/iMPS.py:27-38 (development aid).
usage: python tools/run_imps.py [L] [chi] [cell]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import imps_oracle as io
from temfpy_amd import slater


def ssh(L, t1=-1.5, t2=-1.0):
    M = t1 * np.ones(L - 1); M[1::2] = t2
    M = np.diag(M, 1)
    return M + M.T


L = int(sys.argv[1]) if len(sys.argv) > 1 else 200
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cell = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cut = L // 2
Cs, _ = slater.correlation_matrix(ssh(L)); Cl, _ = slater.correlation_matrix(ssh(L + cell))
warnings.simplefilter("ignore")
ms = slater.C_to_MPS(Cs, {"chi_max": chi}, ortho_center=cut, as_tenpy=False)
n_cell = 2
Cv, _ = slater.correlation_matrix(ssh(L + cell * n_cell))
mv = slater.C_to_MPS(Cv, {"chi_max": chi}, ortho_center=cut, as_tenpy=False)
dense = lambda m: (m.dense_tensors(), [np.asarray(x) for x in m.lam], list(m.form))
Ts, ls, fs = dense(ms); Tv, lv, fv = dense(mv)
for method in ("determinants", "transfer"):
    os.environ["TMF_IMPS"] = method
    for rep in range(2):
        t0 = time.perf_counter()
        res, err = slater.C_to_iMPS(Cs, Cl, {"chi_max": chi}, cell, cut, as_tenpy=False)
        dt = time.perf_counter() - t0
    Tr, lr, fr = io.insert_cells(Ts, ls, fs, res.dense_tensors(), res.lam, cut, n_cell)
    ov = io.overlap(Tv, lv, fv, Tr, lr, fr)
    nr, nv = io.overlap(Tr, lr, fr, Tr, lr, fr).real, io.overlap(Tv, lv, fv, Tv, lv, fv).real
    print(f"{method:13s} L={L} chi={chi} cell={cell}: {dt * 1e3:7.1f} ms, chi of the cell {max(res.chi)}, errors {err.left_unitary:.2e} {err.left_schmidt:.2e}, "
          f"1 - overlap with the direct conversion {abs(abs(ov) / np.sqrt(nr * nv) - 1):.2e}", flush=True)
