"""Probe: Gutzwiller projection of long chains with / without the power-of-two rescaling of the sweeps, against the
numpy oracle run with a renormalisation per step."""
import os, sys, time, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_gpu_gutzwiller import uniform_chain, oracle_inputs
from temfpy_amd import gutzwiller, slater
from oracle import gutzwiller_oracle as gw
warnings.simplefilter("ignore")


def stable(M, cutoff=1e-12):
    M = [np.array(m, complex) for m in M]
    lg = 0.0
    for j in range(len(M) - 1):
        d, cl, cr = M[j].shape
        Q, R = np.linalg.qr(M[j].reshape(d * cl, cr))
        s = np.abs(R).max(); lg += np.log2(s); R = R / s
        M[j] = Q.reshape(d, cl, -1)
        M[j + 1] = np.einsum("ab,pbc->pac", R, M[j + 1])
    B, S, nrm = gw.canonical_form_finite(M, cutoff)
    return S, lg + np.log2(nrm)


for Ls, chi in ((300, 16), (600, 16), (900, 16), (1200, 16), (2200, 16)):
    C, _ = slater.correlation_matrix(uniform_chain(Ls))
    mps = slater.C_to_MPS(C, {"chi_max": chi}, spinful="PH", as_tenpy=False)
    T, q, lam, oc = oracle_inputs(mps)
    M, keep = gw.group_and_project(T, q, lam, oc, "ph")
    S, lg = stable(M)
    t = time.perf_counter(); on = gutzwiller.abrikosov_ph(mps); t_on = time.perf_counter() - t
    dev = lambda r: max(np.abs(np.sort(a)[::-1][:min(len(a), len(b))] - np.sort(b)[::-1][:min(len(a), len(b))]).max()
                        for a, b in zip(r.lam, S))
    print(Ls, chi, "oracle log2", lg, "on", t_on, on.log2_norm, dev(on), flush=True)
    if Ls <= 1200:
        os.environ["TMF_GW_RESCALE"] = "0"
        t = time.perf_counter(); off = gutzwiller.abrikosov_ph(mps); t_off = time.perf_counter() - t
        del os.environ["TMF_GW_RESCALE"]
        print("   off", t_off, np.log2(off.norm), dev(off), flush=True)
