"""One case of soak_gutzwiller.py with the traceback.  usage: python tests/soak/gw_case.py <seed>"""
import os, sys, traceback, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_gutzwiller as tg
from temfpy_amd import gutzwiller, slater
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
L = int(rng.integers(2, 13)); rng_h = float(rng.choice([0.7, 1.5, 3.0])); cplx = bool(rng.integers(0, 2))
x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
H = M[0] + (1j * M[1] if cplx else 0); H = H + H.conj().T
kind = ["ph", "std"][int(rng.integers(0, 2))]; chi = int(rng.choice([16, 64, 256, 4096])); method = ["parallel", "sequential"][int(rng.integers(0, 2))]
print(seed, L, rng_h, cplx, kind, chi, method)
C, _ = slater.correlation_matrix(H, L // 2)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mps = slater.C_to_MPS(C, {"chi_max": chi}, spinful="PH" if kind == "ph" else "simple", as_tenpy=False)
    T, q, lam, oc = tg.oracle_inputs(mps)
    for meth in ("parallel", "sequential"):
        res = (gutzwiller.abrikosov_ph if kind == "ph" else gutzwiller.abrikosov)(mps, method=meth)
        try:
            tg.check(res, T, q, lam, oc, kind, isometry=None if meth == "parallel" else 1e-10)
            print(meth, "ok")
        except Exception:
            print(meth); traceback.print_exc(limit=2)
from oracle import gutzwiller_oracle as gw
Mo, keep = gw.group_and_project(T, q, lam, oc, kind)
B, S, nrm = gw.canonical_form_finite(Mo, 1e-12)
print("norm hip", res.norm, "oracle", nrm, "ratio-1", res.norm / nrm - 1)
print("fermion chi", mps.chi)
