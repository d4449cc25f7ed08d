"""Where the time of the determinant kernel goes: in-kernel cycle stamps of its phases (TMF_PPT_STAMPS=1 diagnostic
path of csrc/det_ppt.hip) on the benchmark workload, next to the launch time from HIP events.
usage: TMF_PPT_STAMPS=1 python tools/ppt_probe.py [L chi reps]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TMF_PPT_STAMPS", "1")
from tests_inputs import random_hopping  # noqa: E402
from temfpy_amd import slater  # noqa: E402
from temfpy_amd.engine import Engine  # noqa: E402
from temfpy_amd.schmidt_utils import to_stopping_condition  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
C, _ = slater.correlation_matrix(random_hopping(L, 0))
eng = Engine("cuda:0")
eng.time_gemm = True
tr = to_stopping_condition({"chi_max": chi})
for _ in range(2):
    eng.run(C, tr, L // 2, L, download=False)
buf = (ctypes.c_uint64 * 32)()
eng.lib.tmf_det_ppt_stamps(buf)
ms = []
for _ in range(reps):
    eng.run(C, tr, L // 2, L, download=False)
    ms.append(eng.kernel_info.det_ms)
eng.lib.tmf_det_ppt_stamps(buf)
v = np.array(list(buf), float)
wg, pairs, nsum = v[16] / reps, v[17] / reps, v[18] / reps
ph = v[:16].reshape(4, 4).mean(axis=0) / reps            # cycles per phase, mean over the four wave slots, per launch
print(f"det launch {np.mean(ms):.3f} ms; {wg:.0f} workgroups, {pairs:.0f} pairs ({pairs / wg:.0f} per workgroup), mean n {nsum / wg:.1f}")
for name, c in zip(("load", "exchange", "tables", "pairs"), ph):
    print(f"  {name:9s} {c / wg:10.0f} cycles per workgroup   {100 * c / ph.sum():5.1f} %")
print(f"  pair phase: {ph[3] / (pairs / 256):.0f} cycles per 64 pairs of one wavefront (4 wavefronts per workgroup)")
sub = v[19:23] / reps
print("  inside the pair phase (all wavefronts): " + ", ".join(f"{n} {100 * c / sub.sum():.1f} %" for n, c in
      zip(("bra side", "masks + sign", "fast determinants + store", "slow queue"), sub)))
h = v[23:31] / reps
print("  pairs by order d of the small determinant: " + ", ".join(f"d={k}{'+' if k == 7 else ''}: {100 * x / h.sum():.2f} %" for k, x in enumerate(h)))
