"""Randomised soak of pfaffian.C_to_MPS (Majorana basis) against the CPU oracle on small random BdG chains: random length,
coupling range, chi_max and orthogonality centre.  Development aid, not part of the tests (it imports the oracle).
usage: python tests/soak/soak_pfaffian.py [cases] [first seed]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pfaffian_oracle as porc  # noqa: E402
from oracle import slater_oracle as orc  # noqa: E402
from temfpy_amd import pfaffian  # noqa: E402


def oracle_dense(cuts, sites):
    T = []
    for i, s in enumerate(sites):
        bra, ket = (cuts[i], cuts[i + 1]) if s.mode == "left" else (cuts[i + 1], cuts[i])
        cb, ck = len(bra.lam), len(ket.lam)
        M = np.zeros((2 * cb, ck), complex)
        for (r0, r1, c0, c1, blk) in s.blocks.values():
            M[s.leg_idx_bra[r0:r1], c0:c1] = blk
        t = M.reshape(2, cb, ck)
        T.append(t if s.mode == "left" else t.transpose(0, 2, 1))
    return T


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = both_raise = 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(2, 15))
    rng_h = float(rng.choice([0.7, 1.5, 3.0]))
    x, y = np.meshgrid(np.arange(2 * L), np.arange(2 * L), indexing="ij")
    M = rng.normal(size=(2 * L, 2 * L)) * np.exp(-abs(x - y) / rng_h)
    H = 1j * (M - M.T)
    chi = int(rng.choice([2, 5, 16, 40, 128]))
    oc = int(rng.integers(1, L)) if (L > 1 and rng.integers(0, 2)) else None
    tag = f"seed {seed}: L={L} range={rng_h} chi={chi} oc={oc}"
    try:
        C = pfaffian.correlation_matrix(H, basis="M->M")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cuts, sites = porc.c_to_mps(C, {"chi_max": chi}, oc)
            mps = pfaffian.C_to_MPS(C, {"chi_max": chi}, basis="M", ortho_center=oc, as_tenpy=False)
        o = oc or L // 2
        event = False
        for b in range(L + 1):
            if len(mps.bonds[b].e) != len(cuts[b].e):
                eo = np.asarray(cuts[b].e if len(cuts[b].e) > len(mps.bonds[b].e) else mps.bonds[b].e)
                near = np.minimum(np.abs(eo - 1e-12), np.abs(1 - eo - 1e-12)).min()
                if near > 1e-13:
                    raise AssertionError(f"bond {b}: {len(mps.bonds[b].e)} vs {len(cuts[b].e)} entangled modes, nearest to the cutoff {near:.1e}")
                event = True
                continue
            if np.abs(np.asarray(mps.bonds[b].e) - np.asarray(cuts[b].e)).max(initial=0) > 1e-12:
                raise AssertionError(f"bond {b}: eigenvalues differ")
            la, lo = np.sort(mps.bonds[b].lam)[::-1], np.sort(cuts[b].lam)[::-1]
            if len(la) != len(lo) or np.abs(la - lo).max() > 1e-8:
                raise AssertionError(f"bond {b}: Schmidt values differ: {len(la)} vs {len(lo)}")
            if (mps.bonds[b].pL + mps.bonds[b].pR) % 2 != (cuts[b].pL + cuts[b].pR) % 2:
                raise AssertionError(f"bond {b}: total parity differs")
        T1, T2 = oracle_dense(cuts, sites), mps.dense_tensors()
        n1 = abs(orc.mps_overlap(T1, cuts[o].lam, T1, cuts[o].lam, o))
        n2 = abs(orc.mps_overlap(T2, mps.lam[o], T2, mps.lam[o], o))
        ov = abs(orc.mps_overlap(T1, cuts[o].lam, T2, mps.lam[o], o)) / np.sqrt(n1 * n2)
        if not event and abs(1 - ov) > 1e-7:
            raise AssertionError(f"state overlap 1 - {1 - ov:.2e}")
    except Exception as e:          # noqa: BLE001
        same = False
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                porc.c_to_mps(pfaffian.correlation_matrix(H, basis="M->M"), {"chi_max": chi}, oc)
        except Exception as e2:     # noqa: BLE001
            same = True
        both_raise += same
        if not same:
            bad += 1
            print("MISMATCH", tag, "->", type(e).__name__, str(e)[:200], flush=True)
print(f"{n_cases} cases, {bad} mismatches, {both_raise} where the oracle raises as well")
sys.exit(1 if bad else 0)
