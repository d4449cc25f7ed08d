"""Phases of the panel QR kernel (csrc/house_slab.hip) inside the Gutzwiller projection of config 5: in-kernel cycle stamps.
usage: TMF_SLAB_REG=0|1 python tools/slab_probe_gw.py [L]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TMF_SLAB_STAMPS", "1")
from tests_inputs import uniform_chain
from temfpy_amd import slater, gutzwiller, _native as nat
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
C, N = slater.correlation_matrix(uniform_chain(L))
mps = slater.C_to_MPS(C, {"chi_max": 512}, as_tenpy=False, spinful="PH")
lib = nat.load()
buf = (ctypes.c_uint64 * 16)()
gutzwiller.abrikosov_ph(mps)
lib.tmf_house_slab_stamps(buf)
res = gutzwiller.abrikosov_ph(mps)
rc = lib.tmf_house_slab_stamps(buf)
if rc != 0:
    print("tmf_house_slab_stamps returned", rc)
v = np.array(list(buf), float)
wg = v[8]
if os.environ.get("TMF_SLAB_REG", "1") != "0":       # the register form with panels stamps other parts
    names = ("load panel", "earlier panels' reflectors", "(loop top)", "owner: column k+1 + its reflector", "owner: other columns",
             "others: apply", "barrier", "R + store")
    print(f"{wg:.0f} blocks of more than 256 rows (register form with panels), mean rows {v[9] / max(wg, 1):.0f}")
    for nm, c in zip(names, v[:8]):
        print(f"  {nm:36s} {c / max(wg, 1):9.0f} ticks per block  {100 * c / max(v[:8].sum(), 1):5.1f} %")
    print({k: round(x * 1e3, 1) for k, x in res.timings.items()})
    sys.exit(0)
names = ("load panel", "reflector blocks: load + barriers", "earlier reflectors: apply", "in-panel steps", "store panel", "R", "Q: apply", "Q over A")
print(f"{wg:.0f} workgroups of the panel kernel, mean rows {v[9] / max(wg, 1):.0f}; {v[:8].sum() / max(wg, 1):.0f} cycles each")
for nm, c in zip(names, v[:8]):
    print(f"  {nm:36s} {c / max(wg, 1):9.0f} cycles per block  {100 * c / max(v[:8].sum(), 1):5.1f} %")
print({k: round(x * 1e3, 1) for k, x in res.timings.items()})
