"""One case of soak_shards.py in detail: which bonds / sites of a range differ from the unsharded conversion, and by how much.
usage: python tests/soak/shard_case.py <seed> [largest L]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.schmidt_utils import to_stopping_condition
seed = int(sys.argv[1]); max_L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = Engine("cuda:0")
rng = np.random.default_rng(seed)
L = int(rng.integers(4, max_L + 1)); rng_h = float(rng.choice([0.7, 1.5, 3.0, 6.0])); cplx = bool(rng.integers(0, 2))
x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
H = M[0] + (1j * M[1] if cplx else 0); H = H + H.conj().T
spinful = [None, None, None, "simple", "PH"][int(rng.integers(0, 5))]
chi = int(rng.choice([8, 32, 128, 300]))
C, _ = slater.correlation_matrix(H)
if spinful:
    C = slater.spinful_correlation_matrix(C, spinful == "PH")
Lf = len(C)
oc = int(rng.integers(1, Lf)) if rng.integers(0, 2) else Lf // 2
nr = int(rng.integers(2, 5)); cuts = sorted(set(rng.integers(1, Lf, size=nr - 1).tolist()))
ranges = list(zip([0] + cuts, cuts + [Lf]))
tr = to_stopping_condition({"chi_max": chi})
print(f"seed {seed}: L={Lf} complex={cplx} spinful={spinful} chi={chi} oc={oc} ranges={ranges}")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    full = eng.run(C, tr, oc, Lf)
    print("full: width", eng.range_width, "iterations", eng.range_iterations_used, "LU fallbacks", full.info.get("lu_fallbacks"))
    for (lo, hi) in ranges:
        part = eng.run(C, tr, oc, Lf, site_range=(lo, hi))
        print(f"range ({lo},{hi}): width", eng.range_width, "iterations", eng.range_iterations_used, "LU fallbacks", part.info.get("lu_fallbacks"))
        for b in range(lo, hi + 1):
            ea, eo = np.asarray(part.bonds[b].e), np.asarray(full.bonds[b].e)
            if ea.shape != eo.shape or not np.array_equal(ea, eo):
                print("  bond", b, "e differs:", (np.abs(ea - eo).max() if ea.shape == eo.shape else (ea.shape, eo.shape)))
            if not np.array_equal(part.bonds[b].lam, full.bonds[b].lam):
                print("  bond", b, "lam differs")
        nbad = 0
        for i in range(lo, hi):
            for bp, bf in zip(part.sites[i].blocks, full.sites[i].blocks):
                if bp[:5] != bf[:5] or not np.array_equal(bp[5], bf[5]):
                    d = np.abs(bp[5] - bf[5]).max() if bp[5].shape == bf[5].shape else -1
                    if nbad < 6:
                        print("  site", i, "block", bp[:5], "differs by", d, " |a|-|b| max", (np.abs(np.abs(bp[5]) - np.abs(bf[5])).max() if d >= 0 else None))
                    nbad += 1
        print("  differing blocks:", nbad)
