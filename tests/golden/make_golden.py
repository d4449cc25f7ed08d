#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE's own NumPy core.

Test infrastructure only -- runs in the build container (needs /root/reference),
never on the GPU box, never imported by the product.

How the reference is loaded: TeMFpy's ``slater.py`` imports
TeNPy at module top and instantiates ``FermionSite()`` at import time
(slater.py:15-16,30).  TeNPy is not installed here, so placeholder module
objects are registered for the *names* touched at import.  No TeNPy behaviour
is emulated: every function that would call into TeNPy (``to_npc_array``,
``networks.mps.MPS``) is simply never called.  All arithmetic recorded below is
executed by the reference's unmodified code:

  SchmidtVectors.from_correlation_matrix  slater.py:702-755 (eigh, block_svd, lowest_sums)
  MPSTensorData.from_schmidt_vectors      slater.py:975-1104
  _tensor_block                           slater.py:828-869

The sector loop of ``to_npc_array`` (slater.py:1132-1141) is replayed with the
bra row-slice of a sector found by popcount equality (``_tensor_block`` asserts
exactly that, slater.py:847-855) instead of through TeNPy's LegPipe, so the
LegPipe row permutation itself is NOT pinned by these fixtures.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
"""
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/src/temfpy"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference sources not present; goldens can only be regenerated in the build container")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Meta(type):  # class-level attribute access (e.g. LegCharge.from_qdict) -> inert callable
        def __getattr__(cls, name):
            return lambda *a, **k: cls()

    class _Placeholder(metaclass=_Meta):  # names only; never used for arithmetic
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Placeholder()

        def __call__(self, *a, **k):
            return _Placeholder()

    npc = mod("tenpy.linalg.np_conserved", Array=_Placeholder, LegCharge=_Placeholder, LegPipe=_Placeholder)
    linalg = mod("tenpy.linalg", np_conserved=npc)
    site = mod("tenpy.networks.site", FermionSite=_Placeholder)
    mps = mod("tenpy.networks.mps", MPS=_Placeholder, TransferMatrix=_Placeholder)
    networks = mod("tenpy.networks", site=site, mps=mps, MPS=_Placeholder)
    mod("tenpy", linalg=linalg, networks=networks)

    pkg = types.ModuleType("temfpy")
    pkg.__path__ = [REF]  # bypass __init__ (needs a hatch-generated _version.py)
    sys.modules["temfpy"] = pkg
    import importlib

    slater = importlib.import_module("temfpy.slater")
    testing = importlib.import_module("temfpy.testing")
    return slater, testing


# ---- inputs (BASELINE.json configs) ---------------------------------
def random_hopping(L, seed, rng_range=3.0):
    """Seeded restatement of src/examples/slater.py:15-20."""
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    scale = np.exp(-abs(x - y) / rng_range)
    M = rng.normal(size=(2, L, L)) * scale
    M = M[0] + 1j * M[1]
    return M + M.T.conj()


def uniform_chain(L, t=-1.0):
    """src/examples/gutzwiller.py:10-12."""
    M = np.diag(t * np.ones(L - 1), 1)
    return M + M.T


def ssh_chain(L, t1=1.0, t2=0.5):
    """Real SSH chain as in src/examples/iMPS.py:6-10."""
    t = np.where(np.arange(L - 1) % 2 == 0, t1, t2)
    M = np.diag(t, 1)
    return M + M.T


CASES = [
    # name, H builder, kwargs for C_to_MPS-equivalent replay
    ("chain_L8_chi8", lambda: uniform_chain(8), dict(chi_max=8)),
    ("chain_L16_chi32", lambda: uniform_chain(16), dict(chi_max=32)),
    ("chain_L32_chi200", lambda: uniform_chain(32), dict(chi_max=200)),  # BASELINE config 1
    ("ssh_L16_chi32", lambda: ssh_chain(16), dict(chi_max=32)),
    ("rand_L8_s0_chi8", lambda: random_hopping(8, 0), dict(chi_max=8)),
    ("rand_L16_s0_chi32", lambda: random_hopping(16, 0), dict(chi_max=32)),
    ("rand_L16_s1_chi200", lambda: random_hopping(16, 1), dict(chi_max=200)),
    ("rand_L32_s0_chi32", lambda: random_hopping(32, 0), dict(chi_max=32)),
    ("rand_L32_s2_chi64", lambda: random_hopping(32, 2), dict(chi_max=64)),
    ("rand_L24_s3_oc7_chi48", lambda: random_hopping(24, 3), dict(chi_max=48, ortho_center=7)),
    ("chainPH_L8_chi64", lambda: uniform_chain(8), dict(chi_max=64, spinful="PH")),
    ("randSimple_L6_s4_chi32", lambda: random_hopping(6, 4), dict(chi_max=32, spinful="simple")),
]


def replay(slater, C, chi_max, ortho_center=None, spinful=None):
    """Replays slater.C_to_MPS (slater.py:1266-1346) collecting dense data."""
    SV, TD = slater.SchmidtVectors, slater.MPSTensorData
    trunc = {"chi_max": chi_max}
    if spinful == "simple":
        C = slater.spinful_correlation_matrix(C, False)
    elif spinful == "PH":
        C = slater.spinful_correlation_matrix(C, True)
    L = len(C)
    oc = ortho_center or L // 2
    out = {"C": C, "L": L, "ortho_center": oc, "chi_max": chi_max}

    def put_bond(b, S):
        m = S.modes
        out[f"b{b}_e"] = m.e
        out[f"b{b}_nfilled"] = np.array([m.n_filled("L"), m.n_filled("R")])
        sets = S.left_sets[:, m.ixL["entangled"]] if S.left_sets is not None else None
        if sets is None:  # right only: recover `sets` from right_sets (slater.py:465)
            sets = np.logical_not(S.right_sets[:, m.ixR["entangled"]][:, ::-1])
        out[f"b{b}_sets"] = sets
        out[f"b{b}_lam_raw"] = S.schmidt_values
        out[f"b{b}_lam"] = S.schmidt_values / np.linalg.norm(S.schmidt_values)  # utils.py:99-103
        keys = np.array(sorted(S.idx_L))
        out[f"b{b}_q"] = keys
        out[f"b{b}_qstart"] = np.array([S.idx_L[k].start for k in keys])
        out[f"b{b}_qstop"] = np.array([S.idx_L[k].stop for k in keys])

    def put_site(i, T):
        out[f"s{i}_det_always"] = np.asarray(T.det_always)
        out[f"s{i}_M"] = T.sometimes_matrix
        out[f"s{i}_sets_bra"] = T.new_sets_bra
        out[f"s{i}_sets_ket"] = T.new_sets_ket
        out[f"s{i}_qtotal"] = np.array(T.qtotal)
        pc_bra = T.new_sets_bra.sum(axis=1)
        qs, r0, r1 = [], [], []
        for q_ket, sl in T.idx_ket.items():  # slater.py:1133-1141
            n_ket = T.new_sets_ket[sl].sum(axis=1)
            assert np.all(n_ket == n_ket[0])
            rows = np.nonzero(pc_bra == n_ket[0])[0]
            if len(rows) == 0:
                continue
            assert np.all(np.diff(rows) == 1)
            blk = T.det_always * slater._tensor_block(
                T.sometimes_matrix, T.new_sets_bra[rows[0] : rows[-1] + 1], T.new_sets_ket[sl]
            )
            out[f"s{i}_blk{q_ket}"] = blk
            qs.append(q_ket), r0.append(rows[0]), r1.append(rows[-1] + 1)
        out[f"s{i}_blkq"] = np.array(qs)
        out[f"s{i}_blkrow0"] = np.array(r0)
        out[f"s{i}_blkrow1"] = np.array(r1)

    Sc = SV.from_correlation_matrix(C, oc, trunc_par=trunc)
    put_bond(oc, Sc)
    S = Sc
    for i in range(oc, L):  # slater.py:1301-1321
        Sn = SV.from_correlation_matrix(C, i + 1, trunc, which="R")
        put_bond(i + 1, Sn)
        put_site(i, TD.from_schmidt_vectors(Sn, S, "right"))
        S = Sn
    S = Sc
    for i in reversed(range(oc)):  # slater.py:1326-1346
        Sn = SV.from_correlation_matrix(C, i, trunc, which="L")
        put_bond(i, Sn)
        put_site(i, TD.from_schmidt_vectors(Sn, S, "left"))
        S = Sn
    return out


def main():
    slater, testing = load_reference()
    warnings.simplefilter("ignore", testing.ComparisonWarning)  # default TEST_ACTION="warn"
    for name, builder, kw in CASES:
        H = builder()
        C, N = slater.correlation_matrix(H)
        data = replay(slater, C, **kw)
        data["H"] = H
        data["C_in"] = C
        data["N"] = np.array(N)
        for k in ("ortho_center", "spinful"):
            if k in kw:
                data["kw_" + k] = np.array(kw[k])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: L={data['L']} N={N} chi@centre={len(data['b%d_lam' % data['ortho_center']])} "
              f"{os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
