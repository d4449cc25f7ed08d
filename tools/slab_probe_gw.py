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
lib.tmf_house_slab_stamps(buf)
v = np.array(list(buf), float)
wg = v[8]
names = ("load panel", "reflector blocks: load + barriers", "earlier reflectors: apply", "in-panel steps", "store panel", "R", "Q: apply", "Q over A")
print(f"{wg:.0f} workgroups of the panel kernel, mean rows {v[9] / max(wg, 1):.0f}; {v[:8].sum() / max(wg, 1):.0f} cycles each")
for nm, c in zip(names, v[:8]):
    print(f"  {nm:36s} {c / max(wg, 1):9.0f} cycles per block  {100 * c / max(v[:8].sum(), 1):5.1f} %")
print({k: round(x * 1e3, 1) for k, x in res.timings.items()})
