"""Throw-away NumPy prototype of the GEMM-rich cut solver that the HIP path implements.

Instead of a full eigh of each diagonal block (slater.py:347) it uses that C is a
projector:  C_RR - C_RR^2 = C_RL C_LR, so the entangled orbitals are the left
singular vectors of the off-diagonal block with sigma^2 = e(1-e) >= cutoff(1-cutoff),
and the filled subspace is the dominant subspace of C_RR minus the entangled part.
Only GEMMs, one tall QR and two <=128-sized Hermitian eigenproblems per cut.
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import slater_oracle as orc


def _rand(shape, rng, cplx):
    r = rng.standard_normal(shape)
    return r + 1j * rng.standard_normal(shape) if cplx else r


def block_subspace(block, off, cutoff, rng, p=64):
    """returns e (desc), U_E (n,k), Qf (n,nf)"""
    n, m = off.shape
    cplx = np.iscomplexobj(block)
    if n == 0:
        return np.zeros(0), np.zeros((0, 0), block.dtype), np.zeros((0, 0), block.dtype)
    if m == 0:
        k = 0
        UE = np.zeros((n, 0), block.dtype)
        e = np.zeros(0)
    else:
        pp = min(p, n, m)
        Y = off @ _rand((m, pp), rng, cplx)
        Q, _ = np.linalg.qr(Y)
        B = Q.conj().T @ off
        Z, sv, _ = np.linalg.svd(B, full_matrices=False)
        s2 = sv**2
        keep = s2 >= cutoff * (1 - cutoff)
        if keep.sum() == pp and pp < min(n, m):
            return block_subspace(block, off, cutoff, rng, p * 2)
        U0 = Q @ Z[:, keep]
        T = U0.conj().T @ block @ U0
        e, X = np.linalg.eigh((T + T.conj().T) / 2)
        e, X = e[::-1], X[:, ::-1]
        UE = U0 @ X
        k = len(e)
    nf = int(round(np.trace(block).real - e.sum()))
    if nf > 0:
        Y = block @ (block @ _rand((n, nf), rng, cplx))
        Y -= UE @ (UE.conj().T @ Y)
        Qf, _ = np.linalg.qr(Y)
        Qf -= UE @ (UE.conj().T @ Qf)
        Qf, _ = np.linalg.qr(Qf)
    else:
        Qf = np.zeros((n, 0), block.dtype)
    return e, UE, Qf


def cut_modes_subspace(C, x, trunc, which, rng):
    cutoff = trunc.svd_min ** 2
    L = len(C)
    nL, nR = x, L - x
    vL = vR = nfL = nfR = None
    if "L" in which:
        eL, UL, QL = block_subspace(C[:x, :x], C[:x, x:], cutoff, rng)
        nfL, k = QL.shape[1], len(eL)
        vL = np.zeros((nL, nL), C.dtype)
        vL[:, :nfL], vL[:, nfL:nfL + k] = QL, UL
        e = eL
    if which == "R":
        eR, UR, QR = block_subspace(C[x:, x:], C[x:, :x], cutoff, rng)
        nfR, k = QR.shape[1], len(eR)
        vR = np.zeros((nR, nR), C.dtype)
        n0 = nR - nfR - k
        vR[:, n0:n0 + k], vR[:, n0 + k:] = UR, QR
        e = 1.0 - eR[::-1]
    elif "R" in which:  # centre: pair right orbitals with the left ones through C_RL
        vRrev = C[x:, :x] @ UL  # = sqrt(e(1-e)) * partner; normalise by the column's own norm
        vRrev /= np.linalg.norm(vRrev, axis=0)  # (1-e is inaccurate in floating point when e -> 1)
        UR = vRrev[:, ::-1].copy()
        UR[:, 1::2] *= -1
        # filled basis of the right block
        blockR = C[x:, x:]
        nfR = int(round(np.trace(blockR).real - (1 - e).sum()))
        cplx = np.iscomplexobj(C)
        Y = blockR @ (blockR @ _rand((nR, nfR), rng, cplx))
        Y -= UR @ (UR.conj().T @ Y)
        QR, _ = np.linalg.qr(Y)
        QR -= UR @ (UR.conj().T @ QR)
        QR, _ = np.linalg.qr(QR)
        vR = np.zeros((nR, nR), C.dtype)
        n0 = nR - nfR - k
        vR[:, n0:n0 + k], vR[:, n0 + k:] = UR, QR
    nferm = int(np.round(np.trace(C).real))
    return orc.Cut(x=x, nL=nL, nR=nR, n_fermion=nferm, e=e, vL=vL, vR=vR, nfL=nfL, nfR=nfR)


def c_to_mps_subspace(C, trunc, ortho_center=None, seed=7):
    trunc = orc.as_trunc(trunc)
    rng = np.random.default_rng(seed)
    saved = orc.cut_modes
    orc.cut_modes = lambda C_, x, t, which="LR": cut_modes_subspace(C_, x, t, which, rng)
    try:
        return orc.c_to_mps(C, trunc, ortho_center)
    finally:
        orc.cut_modes = saved


def compare(C, chi, label):
    import time
    t0 = time.time(); cr, sr = orc.c_to_mps(C, {"chi_max": chi}); t1 = time.time()
    cs, ss = c_to_mps_subspace(C, {"chi_max": chi}); t2 = time.time()
    L = len(C); oc = L // 2
    dS = np.abs(orc.entropies(cr) - orc.entropies(cs)).max()
    same_sets = all(a.sets.shape == b.sets.shape and np.array_equal(a.sets, b.sets) for a, b in zip(cr, cs))
    dlam = max(np.abs(np.sort(a.lam) - np.sort(b.lam)).max() if a.lam.shape == b.lam.shape else 9.0 for a, b in zip(cr, cs))
    same_multiset = all(sorted(map(bytes, a.sets)) == sorted(map(bytes, b.sets)) for a, b in zip(cr, cs))
    label += f" multiset_equal={same_multiset}"
    de = max(np.abs(a.e - b.e).max() for a, b in zip(cr, cs) if a.e.shape == b.e.shape and a.e.size)
    dabs = 0.0
    for a, b in zip(sr, ss):
        for q in a.blocks:
            if q in b.blocks and a.blocks[q][4].shape == b.blocks[q][4].shape:
                dabs = max(dabs, np.abs(np.abs(a.blocks[q][4]) - np.abs(b.blocks[q][4])).max())
    Tr, Ts = orc.dense_tensors(cr, sr), orc.dense_tensors(cs, ss)
    ov = None
    if max(len(c.lam) for c in cr) <= 200:
        ov = abs(orc.mps_overlap(Tr, cr[oc].lam, Ts, cs[oc].lam, oc)) / abs(orc.mps_overlap(Tr, cr[oc].lam, Tr, cr[oc].lam, oc))
    print(f"{label}: sets_equal={same_sets} max|dS|={dS:.2e} max|dlam|={dlam:.2e} max|de|={de:.2e} "
          f"max||B|-|B'||={dabs:.2e} 1-overlap={-1 if ov is None else 1-ov:.2e}  t_ref={t1-t0:.1f}s t_sub={t2-t1:.1f}s")


if __name__ == "__main__":
    sys.path.insert(0, "tests/golden")
    from make_golden import random_hopping, uniform_chain, ssh_chain
    for L, seed, chi in [(16, 0, 32), (32, 0, 64), (32, 2, 200), (64, 1, 64), (96, 0, 128)]:
        C, _ = orc.correlation_matrix(random_hopping(L, seed))
        compare(C, chi, f"rand L={L} s={seed} chi={chi}")
    for L, chi in [(16, 32), (32, 200), (64, 64)]:
        C, _ = orc.correlation_matrix(uniform_chain(L))
        compare(C, chi, f"chain L={L} chi={chi}")
    C, _ = orc.correlation_matrix(ssh_chain(32))
    compare(C, 64, "ssh L=32 chi=64")
    C, _ = orc.correlation_matrix(uniform_chain(16))
    compare(orc.spinful_correlation_matrix(C, True), 64, "chainPH L=16 chi=64")
