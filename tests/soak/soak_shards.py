"""Randomised soak of the site sharding: random inputs (length, filling, range, chi, spinful mode, orthogonality centre) cut
into random contiguous site ranges, every range converted on its own and compared BITWISE with the unsharded conversion
(tensor blocks, Schmidt values, occupation masks).  One GPU, no process group: every range decides the range-finder width on
its own, so cases in which a shard would decide differently from the whole chain are reported separately (the multi-GPU
path reduces those decisions over the ranks).  Development aid.
usage: python tests/soak/soak_shards.py [cases] [first seed] [largest L]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from temfpy_amd import slater  # noqa: E402
from temfpy_amd.engine import Engine  # noqa: E402
from temfpy_amd.schmidt_utils import to_stopping_condition  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_L = int(sys.argv[3]) if len(sys.argv) > 3 else 64
eng = Engine("cuda:0")
bad = decided = 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(4, max_L + 1))
    rng_h = float(rng.choice([0.7, 1.5, 3.0, 6.0]))
    cplx = bool(rng.integers(0, 2))
    x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    M = rng.normal(size=(2, L, L)) * np.exp(-abs(x - y) / rng_h)
    H = M[0] + (1j * M[1] if cplx else 0)
    H = H + H.conj().T
    spinful = [None, None, None, "simple", "PH"][int(rng.integers(0, 5))]
    chi = int(rng.choice([8, 32, 128, 300]))
    C, _ = slater.correlation_matrix(H)
    if spinful:
        C = slater.spinful_correlation_matrix(C, spinful == "PH")
    Lf = len(C)
    oc = int(rng.integers(1, Lf)) if rng.integers(0, 2) else Lf // 2
    nr = int(rng.integers(2, 5))
    cuts = sorted(set(rng.integers(1, Lf, size=nr - 1).tolist()))
    ranges = list(zip([0] + cuts, cuts + [Lf]))
    tr = to_stopping_condition({"chi_max": chi})
    tag = f"seed {seed}: L={Lf} complex={cplx} spinful={spinful} chi={chi} oc={oc} ranges={ranges}"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            full = eng.run(C, tr, oc, Lf)
            wf, itf = eng.range_width, eng.range_iterations_used
            for (lo, hi) in ranges:
                part = eng.run(C, tr, oc, Lf, site_range=(lo, hi))
                if (eng.range_width, eng.range_iterations_used) != (wf, itf):
                    decided += 1
                    break
                for i in range(lo, hi):
                    if len(part.sites[i].blocks) != len(full.sites[i].blocks):
                        raise AssertionError(f"site {i}: number of blocks")
                    for bp, bf in zip(part.sites[i].blocks, full.sites[i].blocks):
                        if bp[:5] != bf[:5] or not np.array_equal(bp[5], bf[5]):
                            raise AssertionError(f"site {i}: block {bp[:5]} differs by {np.abs(bp[5] - bf[5]).max() if bp[5].shape == bf[5].shape else 'shape'}")
                for b in range(lo, hi + 1):
                    if not np.array_equal(part.bonds[b].lam, full.bonds[b].lam) or not np.array_equal(part.bonds[b].masks, full.bonds[b].masks):
                        raise AssertionError(f"bond {b}: Schmidt values / masks differ")
    except Exception as e:          # noqa: BLE001
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                eng.run(C, tr, oc, Lf)
            bad += 1
            print("MISMATCH", tag, "->", type(e).__name__, str(e)[:200], flush=True)
        except Exception:           # noqa: BLE001   the unsharded conversion raises as well (e.g. a singular always-block)
            pass
print(f"{n_cases} cases, {bad} mismatches, {decided} where a shard alone picks another range-finder setting")
sys.exit(1 if bad else 0)
