"""O(d) evaluation of the sign: verify the table/mask formulas against brute force."""
import numpy as np
rng = np.random.default_rng(2)

def exchange(M, pivots):
    G = M.astype(complex).copy()
    for r, c in pivots:
        p = G[r, c]
        row = G[r, :].copy(); col = G[:, c].copy()
        G = G - np.outer(col, row) / p
        G[r, :] = -row / p
        G[:, c] = col / p
        G[r, c] = 1.0 / p
    return G

def inv_parity(seq):
    seq = list(seq); inv = 0
    for i in range(len(seq)):
        for j in range(i + 1, len(seq)):
            inv += seq[i] > seq[j]
    return inv

popc = lambda x: bin(x).count("1")
bad = 0
for trial in range(4000):
    sb, sk = rng.integers(3, 12), rng.integers(3, 12)
    n = rng.integers(1, min(sb, sk) + 1)
    M = rng.standard_normal((sb, sk)) + 1j * rng.standard_normal((sb, sk))
    # pivots in an arbitrary order with arbitrary pairing
    prow_seq = list(rng.permutation(rng.choice(sb, n, replace=False)))
    pcol_seq = list(rng.permutation(rng.choice(sk, n, replace=False)))
    pivots = list(zip(prow_seq, pcol_seq))
    G = exchange(M, pivots)
    prod_piv = 1.0
    Gtmp = M.astype(complex).copy()
    for r, c in pivots:
        prod_piv *= Gtmp[r, c]
        Gtmp = exchange(Gtmp, [(r, c)])
    row_of = {c: r for r, c in pivots}; col_of = {r: c for r, c in pivots}
    PA = sum(1 << r for r in prow_seq); PB = sum(1 << c for c in pcol_seq)
    NPB = ((1 << sk) - 1) & ~PB
    # sector constants
    astar = sorted(prow_seq)
    # det M*_sorted = sign * prod_piv, sign = parity(inv of pivot-row order) + parity(inv of pivot-col order)
    c_sector = (inv_parity(prow_seq) + inv_parity(pcol_seq)) & 1
    # tables: invr[r] = inversions (col_of sequence over pivot rows ascending) involving r
    seq = [col_of[r] for r in astar]
    invr = {}
    for i, r in enumerate(astar):
        invr[r] = sum(seq[j] > seq[i] for j in range(i)) + sum(seq[j] < seq[i] for j in range(i + 1, n))
    a = sorted(rng.choice(sb, n, replace=False)); b = sorted(rng.choice(sk, n, replace=False))
    true = np.linalg.det(M[np.ix_(a, b)])
    am = sum(1 << r for r in a); bm = sum(1 << c for c in b)
    A_in = [r for r in range(sb) if (am & ~PA) >> r & 1]; A_out = [r for r in range(sb) if (PA & ~am) >> r & 1]
    B_in = [c for c in range(sk) if (bm & ~PB) >> c & 1]; B_out = [c for c in range(sk) if (PB & ~bm) >> c & 1]
    da, db = len(A_in), len(B_in)
    Rm = (am & ~PA) | sum(1 << row_of[c] for c in B_out)
    Cm = (bm & ~PB) | sum(1 << col_of[r] for r in A_out)
    Rs = [r for r in range(sb) if Rm >> r & 1]; Cs = [c for c in range(sk) if Cm >> c & 1]
    d = len(Rs)
    small = np.linalg.det(G[np.ix_(Rs, Cs)]) if d else 1.0
    par = c_sector
    par += sum(B_in) + sum(B_out)                                   # T1
    par += d * (sk - 1) + sum(Cs) + d * (d - 1) // 2                # T5
    # tau terms
    par += sum(popc(NPB >> (c1 + 1)) - sum(c0 > c1 for c0 in B_in) for c1 in B_out)          # I_X0X1
    par += sum(popc(NPB >> (col_of[r] + 1)) for r in A_out) + sum(popc(PB & ((1 << c0) - 1)) for c0 in B_in) \
        + sum(col_of[r] < c0 for r in A_out for c0 in B_in)                                   # I_X0Y1 - T_full
    par += inv_parity([row_of[c] for c in B_out])                                              # I_X1X1
    par += sum(r < row_of[c] for c in B_out for r in A_in)                                     # I_X1Y0
    par += db * (n - da)                                                                        # I_X1Y1
    par += sum(popc(PA >> (r0 + 1)) - sum(r1 > r0 for r1 in A_out) for r0 in A_in)            # I_Y0Y1
    par += sum(invr[r] for r in A_out) + inv_parity([col_of[r] for r in A_out])               # I_Y1Y1 - inv_PP
    pred = (-1) ** (par & 1) * prod_piv * small
    if abs(pred - true) > 1e-8 * max(1, abs(true)):
        bad += 1
print("mismatches:", bad, "of 4000")
