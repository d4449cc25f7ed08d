"""Phases of the Householder slab QR kernel (csrc/house_slab.hip), in-kernel cycle stamps.  usage: python tools/slab_probe.py [world rank]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TMF_SLAB_STAMPS", "1")
import torch
from tests_inputs import random_hopping
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.multi_gpu import shard_sites
from temfpy_amd.schmidt_utils import to_stopping_condition
L = 1024
C, _ = slater.correlation_matrix(random_hopping(L, 0))
d_C = torch.from_numpy(np.ascontiguousarray(C).reshape(-1)).to("cuda:0")
tr = to_stopping_condition({"chi_max": 512})
rng = shard_sites(L, L // 2, int(sys.argv[1]))[int(sys.argv[2])] if len(sys.argv) > 2 else (0, L)
eng = Engine("cuda:0")
buf = (ctypes.c_uint64 * 16)()
for _ in range(2):
    eng.run(d_C, tr, L // 2, L, download=False, site_range=rng)
eng.lib.tmf_house_slab_stamps(buf)
reps = 5
for _ in range(reps):
    eng.run(d_C, tr, L // 2, L, download=False, site_range=rng)
eng.lib.tmf_house_slab_stamps(buf)
v = np.array(list(buf), float)
wg = v[8]
names = ("load panel", "reflector blocks: load + barriers", "earlier reflectors: apply", "in-panel steps", "store panel", "R", "Q: apply", "Q over A")
print(f"range {rng}: {wg / reps:.0f} slabs per conversion (both launches), mean rows {v[9] / wg:.0f}; {v[:8].sum() / wg:.0f} cycles per slab")
for nm, c in zip(names, v[:8]):
    print(f"  {nm:36s} {c / wg:9.0f} cycles per slab  {100 * c / v[:8].sum():5.1f} %")
