#!/usr/bin/env python3
"""Full-size reference summaries: the REFERENCE's own NumPy core at the BASELINE sizes.

Test infrastructure only -- runs in the build container (needs /root/reference; loader and
caveats in make_golden.py), never on the GPU box, never imported by the product.

The dense tensors of a full-size conversion are 1.5 GB, so what is kept per case is what a
gauge transformation cannot change (eigenvectors are defined up to a phase, so tensors are
equal only up to a diagonal unitary on every bond):

  per bond   chi, the normalised Schmidt values (float64), S(b), the unnormalised norm,
             n_filled L/R, the entangled eigenvalues e, the occupation patterns (``sets`` packed
             little-endian, rows in the reference's order; also their SHA-1) and SHA-1 of q_left
  per site   Frobenius norm of every charge block (det_always included, slater.py:1137-1141),
             |det_always|, and the 2-norm of every row of the merged (p, bra) leg (float32)

Replays the loop of slater.C_to_MPS (slater.py:1293-1346) exactly as make_golden.replay does.

Pfaffian cases (BASELINE config 4): the cut decomposition only - pfaffian.SchmidtVectors.from_correlation_matrix
(pfaffian.py:685-920, :1008-1248) for every bond, replayed as pfaffian.C_to_MPS walks the chain (pfaffian.py:1832-1914).
No sub-Pfaffian is evaluated, so the third-party pfapack (absent here) is not involved: per bond chi, e, the
vacuum parities pL / pR, normalised Schmidt values, S(b), the occupation patterns (packed + SHA-1, rows in the
reference's (parity, number) order) and the sector boundaries idx_n.

Usage:  python tests/golden/make_golden_summary.py [case ...]     (writes tests/golden/full/*.npz)
"""
import hashlib
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, random_hopping, uniform_chain  # noqa: E402

OUT = os.path.join(HERE, "full")

CASES = {
    # BASELINE config 2 / config 3, config 5's Slater stage (src/examples/gutzwiller.py:15-23)
    "cfg2_rand_L256_s0_chi128": (lambda: random_hopping(256, 0), dict(chi_max=128)),
    "cfg3_rand_L1024_s0_chi512": (lambda: random_hopping(1024, 0), dict(chi_max=512)),
    "cfg5_chainPH_L512_chi512": (lambda: uniform_chain(512), dict(chi_max=512, spinful="PH")),
    "rand_L1024_s1_chi512": (lambda: random_hopping(1024, 1), dict(chi_max=512)),
}


def sha(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def summarise(slater, C, chi_max, ortho_center=None, spinful=None, log=None):
    SV, TD = slater.SchmidtVectors, slater.MPSTensorData
    trunc = {"chi_max": chi_max}
    if spinful == "simple":
        C = slater.spinful_correlation_matrix(C, False)
    elif spinful == "PH":
        C = slater.spinful_correlation_matrix(C, True)
    L = len(C)
    oc = ortho_center or L // 2
    chi = np.zeros(L + 1, np.int64)
    S = np.zeros(L + 1)
    nrm = np.zeros(L + 1)
    nfill = np.zeros((L + 1, 2), np.int64)
    lam, e, h_sets, h_q = [None] * (L + 1), [None] * (L + 1), np.zeros((L + 1, 20), np.uint8), np.zeros((L + 1, 20), np.uint8)
    blkq, blkn, rown, deta = [None] * L, [None] * L, [None] * L, np.zeros(L)
    packed = [None] * (L + 1)

    def put_bond(b, V):
        m = V.modes
        sets = V.left_sets[:, m.ixL["entangled"]] if V.left_sets is not None else None
        if sets is None:  # slater.py:465
            sets = np.logical_not(V.right_sets[:, m.ixR["entangled"]][:, ::-1])
        sv = V.schmidt_values
        nrm[b] = np.linalg.norm(sv)
        lam[b] = sv / nrm[b]  # utils.py:99-103
        p = lam[b] ** 2
        S[b] = -(p[p > 0] * np.log(p[p > 0])).sum()
        chi[b] = len(sv)
        e[b] = np.asarray(m.e, np.float64)
        nfill[b] = m.n_filled("L"), m.n_filled("R")
        packed[b] = np.packbits(np.asarray(sets, bool), axis=1, bitorder="little").reshape(-1)
        h_sets[b] = sha(packed[b])
        q = np.zeros(len(sv), np.int64)
        for k, sl in V.idx_L.items():
            q[sl] = k
        h_q[b] = sha(q)

    def put_site(i, T):
        pc_bra = T.new_sets_bra.sum(axis=1)
        qs, ns = [], []
        rn2 = np.zeros(len(T.new_sets_bra))
        for q_ket, sl in T.idx_ket.items():  # slater.py:1133-1141
            n_ket = T.new_sets_ket[sl].sum(axis=1)
            rows = np.nonzero(pc_bra == n_ket[0])[0]
            if len(rows) == 0:
                continue
            blk = T.det_always * slater._tensor_block(T.sometimes_matrix, T.new_sets_bra[rows[0]: rows[-1] + 1],
                                                      T.new_sets_ket[sl])
            a2 = np.abs(blk) ** 2
            qs.append(q_ket), ns.append(np.sqrt(a2.sum()))
            rn2[rows[0]: rows[-1] + 1] += a2.sum(axis=1)
        blkq[i], blkn[i] = np.array(qs, np.int64), np.array(ns)
        rown[i] = np.sqrt(rn2).astype(np.float32)
        deta[i] = abs(T.det_always)

    t0 = time.time()
    Sc = SV.from_correlation_matrix(C, oc, trunc_par=trunc)
    put_bond(oc, Sc)
    V = Sc
    for i in range(oc, L):  # slater.py:1301-1321
        Vn = SV.from_correlation_matrix(C, i + 1, trunc, which="R")
        put_bond(i + 1, Vn)
        put_site(i, TD.from_schmidt_vectors(Vn, V, "right"))
        V = Vn
        if log and i % 64 == 0:
            log(f"  site {i} ({time.time() - t0:.0f} s)")
    V = Sc
    for i in reversed(range(oc)):  # slater.py:1326-1346
        Vn = SV.from_correlation_matrix(C, i, trunc, which="L")
        put_bond(i, Vn)
        put_site(i, TD.from_schmidt_vectors(Vn, V, "left"))
        V = Vn
        if log and i % 64 == 0:
            log(f"  site {i} ({time.time() - t0:.0f} s)")
    wall = time.time() - t0

    def flat(lst, dt):
        off = np.concatenate(([0], np.cumsum([len(x) for x in lst]))).astype(np.int64)
        return np.concatenate([np.asarray(x, dt) for x in lst]) if lst else np.zeros(0, dt), off

    lam_f, lam_off = flat(lam, np.float64)
    e_f, e_off = flat(e, np.float64)
    bq_f, b_off = flat(blkq, np.int64)
    bn_f, _ = flat(blkn, np.float64)
    rn_f, r_off = flat(rown, np.float32)
    pk_f, pk_off = flat(packed, np.uint8)
    return dict(L=L, ortho_center=oc, chi_max=chi_max, chi=chi, S=S, lam_norm=nrm, n_filled=nfill, lam=lam_f, lam_off=lam_off,
                e=e_f, e_off=e_off, sets_packed=pk_f, sets_off=pk_off, sets_sha1=h_sets, q_sha1=h_q, blk_q=bq_f, blk_norm=bn_f, blk_off=b_off,
                row_norm=rn_f, row_off=r_off, abs_det_always=deta, reference_wall_s=np.array(wall),
                reference_cores=np.array(len(os.sched_getaffinity(0))))


PF_CASES = {
    # BASELINE config 4: Majorana-basis Kitaev chain (src/examples/iMPS_pfaffian.py:7-11) and the random BdG
    # chain (src/examples/pfaffian.py:13-17, seeded) at L = 512, chi_max = 256
    "cfg4_kitaev_L512_chi256": (lambda g: g.kitaev_majorana_H(512, 1.5j, 1j), dict(chi_max=256)),
    "cfg4_randbdg_L512_s0_chi256": (lambda g: g.random_majorana_H(512, 0), dict(chi_max=256)),
    "pf_randbdg_L64_s5_chi64": (lambda g: g.random_majorana_H(64, 5), dict(chi_max=64)),
}


def summarise_pfaffian(pf, C, chi_max, ortho_center=None, log=None):
    SV = pf.SchmidtVectors
    trunc = {"chi_max": chi_max}
    L = len(C) // 2
    oc = ortho_center or L // 2
    chi, S, nrm = np.zeros(L + 1, np.int64), np.zeros(L + 1), np.zeros(L + 1)
    par = np.full((L + 1, 2), -1, np.int64)
    lam, e, packed, idxn = [None] * (L + 1), [None] * (L + 1), [None] * (L + 1), [None] * (L + 1)
    h_sets = np.zeros((L + 1, 20), np.uint8)

    def put_bond(b, V):
        m = V.modes
        sets = V.left_sets if V.left_sets is not None else V.right_sets[:, ::-1]
        sv = V.schmidt_values
        nrm[b] = np.linalg.norm(sv)
        lam[b] = sv / nrm[b]
        p = lam[b] ** 2
        S[b] = -(p[p > 0] * np.log(p[p > 0])).sum()
        chi[b] = len(sv)
        e[b] = np.asarray(m.e, np.float64)
        par[b] = (-1 if m.pL is None else m.pL, -1 if m.pR is None else m.pR)
        packed[b] = np.packbits(np.asarray(sets, bool), axis=1, bitorder="little").reshape(-1)
        h_sets[b] = sha(packed[b])
        ks = sorted(V.idx_n)
        idxn[b] = np.array([[k, V.idx_n[k].start, V.idx_n[k].stop] for k in ks], np.int64).reshape(-1)

    t0 = time.time()
    Sc = SV.from_correlation_matrix(C, oc, trunc, basis="M")
    put_bond(oc, Sc)
    parity = Sc.parity()
    for i in range(oc, L):  # pfaffian.py:1861-1881
        put_bond(i + 1, SV.from_correlation_matrix(C, i + 1, trunc, which="R", basis="M", total_parity=parity))
        if log and i % 64 == 0:
            log(f"  bond {i + 1} ({time.time() - t0:.0f} s)")
    for i in reversed(range(oc)):  # pfaffian.py:1887-1907
        put_bond(i, SV.from_correlation_matrix(C, i, trunc, which="L", basis="M", total_parity=parity))
        if log and i % 64 == 0:
            log(f"  bond {i} ({time.time() - t0:.0f} s)")
    wall = time.time() - t0

    def flat(lst, dt):
        off = np.concatenate(([0], np.cumsum([len(x) for x in lst]))).astype(np.int64)
        return np.concatenate([np.asarray(x, dt) for x in lst]), off

    lam_f, lam_off = flat(lam, np.float64)
    e_f, e_off = flat(e, np.float64)
    pk_f, pk_off = flat(packed, np.uint8)
    in_f, in_off = flat(idxn, np.int64)
    return dict(L=L, ortho_center=oc, chi_max=chi_max, chi=chi, S=S, lam_norm=nrm, parities=par, total_parity=np.array(parity),
                lam=lam_f, lam_off=lam_off, e=e_f, e_off=e_off, sets_packed=pk_f, sets_off=pk_off, sets_sha1=h_sets,
                idx_n=in_f, idx_n_off=in_off, reference_wall_s=np.array(wall),
                reference_cores=np.array(len(os.sched_getaffinity(0))))


def main_pfaffian(names):
    import make_golden_pfaffian as g
    pf, testing = g.load_reference_pfaffian()
    warnings.simplefilter("ignore")
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        builder, kw = PF_CASES[name]
        C = pf.correlation_matrix(builder(g), "M->M")
        print(f"{name}: running the reference's cut decomposition ...", flush=True)
        data = summarise_pfaffian(pf, C, log=lambda s: print(s, flush=True), **kw)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        oc = int(data["ortho_center"])
        print(f"{name}: L={data['L']} parity={int(data['total_parity'])} chi@centre={data['chi'][oc]} "
              f"S(centre)={data['S'][oc]:.9f} reference wall {float(data['reference_wall_s']):.1f} s, "
              f"{os.path.getsize(path) / 1e6:.2f} MB", flush=True)


def main():
    names = sys.argv[1:] or list(CASES)
    pf_names = [n for n in names if n in PF_CASES]
    if pf_names:
        main_pfaffian(pf_names)
    names = [n for n in names if n not in PF_CASES]
    if not names:
        return
    slater, testing = load_reference()
    warnings.simplefilter("ignore", testing.ComparisonWarning)
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        builder, kw = CASES[name]
        H = builder()
        C, N = slater.correlation_matrix(H)
        print(f"{name}: running the reference core ...", flush=True)
        data = summarise(slater, C, log=lambda s: print(s, flush=True), **kw)
        data["N"] = np.array(N)
        if "spinful" in kw:
            data["kw_spinful"] = np.array(kw["spinful"])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        oc = int(data["ortho_center"])
        print(f"{name}: L={data['L']} N={N} chi@centre={data['chi'][oc]} S(centre)={data['S'][oc]:.9f} "
              f"reference wall {float(data['reference_wall_s']):.1f} s, {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    main()
