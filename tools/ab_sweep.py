"""A/B check of the two host drivers of the sweep: Engine.run through the C++ staged ABI (csrc/sweep.cpp) against the
Python orchestration it replaces (Engine.run_gen).  Same kernels, same descriptors -> the results must agree bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping, uniform_chain  # noqa: E402
from temfpy_amd import slater  # noqa: E402
from temfpy_amd.engine import Engine  # noqa: E402
from temfpy_amd.schmidt_utils import to_stopping_condition  # noqa: E402


def compare(a, b, tag):
    bad = []
    L = a.L
    for x in range(L + 1):
        p, q = a.bonds[x], b.bonds[x]
        if (p is None) != (q is None):
            bad.append(f"bond {x} presence")
            continue
        if p is None:
            continue
        for f in ("e", "masks", "lam_raw", "q_left"):
            if not np.array_equal(getattr(p, f), getattr(q, f)):
                bad.append(f"bond {x} {f} max diff {np.abs(np.asarray(getattr(p, f), float) - np.asarray(getattr(q, f), float)).max() if getattr(p, f).shape == getattr(q, f).shape else 'shape'}")
        if (p.n_filled_left, p.n_filled_right) != (q.n_filled_left, q.n_filled_right):
            bad.append(f"bond {x} filled")
    if len(a.sites):
        for i in range(L):
            s, t = a.sites[i], b.sites[i]
            if (s is None) != (t is None):
                bad.append(f"site {i} presence")
                continue
            if s is None:
                continue
            if s.det_always != t.det_always:
                bad.append(f"site {i} det {s.det_always} {t.det_always}")
            if len(s.blocks) != len(t.blocks):
                bad.append(f"site {i} nblocks")
                continue
            for u, v in zip(s.blocks, t.blocks):
                if u[:5] != v[:5] or not np.array_equal(u[5], v[5]):
                    bad.append(f"site {i} block q={u[0]} diff {np.abs(u[5] - v[5]).max() if u[5].shape == v[5].shape else 'shape'}")
            if not (np.array_equal(s.bra_p, t.bra_p) and np.array_equal(s.bra_alpha, t.bra_alpha)):
                bad.append(f"site {i} merged leg")
    print(f"{tag}: {'IDENTICAL' if not bad else 'DIFFERS: ' + '; '.join(bad[:6])}", flush=True)
    return not bad


def main():
    py, cpp = Engine("cuda:0"), Engine("cuda:0")
    py.sweep_impl, cpp.sweep_impl = "python", "cpp"
    cpp.lu_method = "single"       # the Python orchestration uses the one-workgroup-per-site LU kernel
    cpp.filled_blocks = 16         # ... and 16-column outer blocks in the filled-basis Gram-Schmidt
    blk = Engine("cuda:0")         # C++ path with the fully pivoted blocked LU over several launches
    blk.lu_method = "blocked"
    loc = Engine("cuda:0")         # default: pivoting inside the diagonal blocks
    fb = Engine("cuda:0")          # ... followed by the forced fallback (must equal the pivoted path bit for bit)
    fb.lu_method = "fallback"
    cases = [("rand L=16 chi=32", slater.correlation_matrix(random_hopping(16, 0))[0], dict(chi_max=32), {}),
             ("rand L=48 chi=32 oc=7", slater.correlation_matrix(random_hopping(48, 5))[0], dict(chi_max=32), dict(oc=7)),
             ("chain real L=40 chi=64", slater.correlation_matrix(uniform_chain(40) + np.diag(0.3 * np.cos(1.7 * np.arange(40))))[0], dict(chi_max=64), {}),
             ("chain L=32 chi=200", slater.correlation_matrix(uniform_chain(32))[0], dict(chi_max=200), {}),
             ("PH chain L=24", slater.spinful_correlation_matrix(slater.correlation_matrix(uniform_chain(24))[0], True), dict(chi_max=64), {}),
             ("rand L=20 unlimited chi", slater.correlation_matrix(random_hopping(20, 2))[0], dict(chi_max=None), {}),
             ("rand L=24 sectors", slater.correlation_matrix(random_hopping(24, 3))[0], dict(chi_max=16, sectors=[q for q in range(0, 25) if q != 6]), {}),
             ("rand L=96 shard (20,61)", slater.correlation_matrix(random_hopping(96, 1))[0], dict(chi_max=48), dict(rng=(20, 61))),
             ("rand L=96 shard (48,96)", slater.correlation_matrix(random_hopping(96, 1))[0], dict(chi_max=48), dict(rng=(48, 96))),
             ("dense L=160 (rank > 64)", slater.correlation_matrix(np.random.default_rng(3).normal(size=(160, 160)) + 0)[0] if False else None, None, None),
             ("rand L=256 chi=128", slater.correlation_matrix(random_hopping(256, 0))[0], dict(chi_max=128), {}),
             ("rand L=1024 chi=512", slater.correlation_matrix(random_hopping(1024, 0))[0], dict(chi_max=512), {})]
    rng = np.random.default_rng(11)
    M = rng.normal(size=(160, 160))
    cases[9] = ("dense L=160 (rank > 64)", slater.correlation_matrix(M + M.T)[0], dict(chi_max=32), {})
    ok = True
    for tag, C, tr, kw in cases:
        tr = to_stopping_condition(tr)
        L = len(C)
        oc = kw.get("oc") or L // 2
        try:
            a = py.run(C, tr, oc, L, site_range=kw.get("rng"))
            b = cpp.run(C, tr, oc, L, site_range=kw.get("rng"))
        except Exception as exc:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            print(f"{tag}: EXCEPTION {type(exc).__name__}: {exc}", flush=True)
            ok = False
            continue
        ok &= compare(a, b, tag)
        try:
            d = blk.run(C, tr, oc, L, site_range=kw.get("rng"))
            worst = 0.0
            for i in range(L):
                if b.sites[i] is None:
                    continue
                for u, v in zip(b.sites[i].blocks, d.sites[i].blocks):
                    assert u[:5] == v[:5]
                    worst = max(worst, np.abs(u[5] - v[5]).max() / max(np.abs(u[5]).max(), 1e-300))
            print(f"   pivoted blocked LU, 64-column Gram-Schmidt blocks vs the above: max relative block deviation {worst:.2e}", flush=True)
            e = loc.run(C, tr, oc, L, site_range=kw.get("rng"))
            loc._stage_timings()
            minp = loc.kernel_info.lu_min_pivot
            worst = 0.0
            for i in range(L):
                if d.sites[i] is None:
                    continue
                worst = max(worst, abs(d.sites[i].det_always - e.sites[i].det_always) / abs(d.sites[i].det_always))
                for u, v in zip(d.sites[i].blocks, e.sites[i].blocks):
                    assert u[:5] == v[:5]
                    worst = max(worst, np.abs(u[5] - v[5]).max() / max(np.abs(u[5]).max(), 1e-300))
            print(f"   block-local pivoting vs fully pivoted: max relative deviation {worst:.2e}; smallest pivot {minp:.3e}, largest "
                  f"|D^-1| {loc.kernel_info.lu_max_inverse:.3e}; fallbacks {loc.kernel_info.lu_fallbacks}", flush=True)
            ok &= worst < 1e-9
            f = fb.run(C, tr, oc, L, site_range=kw.get("rng"))
            fb._stage_timings()
            same = all(d.sites[i] is None or (d.sites[i].det_always == f.sites[i].det_always and all(
                np.array_equal(u[5], v[5]) for u, v in zip(d.sites[i].blocks, f.sites[i].blocks))) for i in range(L))
            print(f"   forced fallback vs fully pivoted: {'IDENTICAL' if same else 'DIFFERS'} (fallbacks {fb.kernel_info.lu_fallbacks})", flush=True)
            ok &= same
        except Exception as exc:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            ok = False
        if a.info["checks"] != b.info["checks"]:
            print("   checks differ:", a.info["checks"], b.info["checks"])
    # timing at full size
    import torch
    C = cases[-1][1]
    tr = to_stopping_condition({"chi_max": 512})
    for name, eng in (("python", py), ("cpp single-kernel LU", cpp), ("cpp blocked LU", blk)):
        for _ in range(3):
            eng.run(C, tr, 512, 1024)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m = eng.run(C, tr, 512, 1024, download=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name}: {dt * 1e3:.2f} ms per conversion (device-resident), stages {dict((k, round(v * 1e3, 2)) for k, v in m.timings.items())}", flush=True)
    print("ALL IDENTICAL" if ok else "SOME DIFFER")


if __name__ == "__main__":
    main()
