"""Per-step timeline of the host -> host pipeline of bench.py (download of conversion k overlapped with conversion k + 1):
step period, time inside Engine.run, time blocked in the wait for conversion k - 2."""
import os
import sys
import time
import gc

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping  # noqa: E402
from temfpy_amd import slater  # noqa: E402
from temfpy_amd.engine import Engine  # noqa: E402
from temfpy_amd.schmidt_utils import to_stopping_condition  # noqa: E402

L, chi, n = 1024, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 30
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
C, _ = slater.correlation_matrix(random_hopping(L, 0))
tr = to_stopping_condition({"chi_max": chi})
eng = Engine("cuda:0")
res = []
for _ in range(3):
    eng.run(C, tr, L // 2, L, download=True)
gc.collect()
gc.disable()
t_prev = time.perf_counter()
rows = []
for k in range(n):
    t0 = time.perf_counter()
    m = eng.run(C, tr, L // 2, L, download="async")
    t1 = time.perf_counter()
    res.append(m)
    if len(res) > depth:
        res.pop(0).wait()
    t2 = time.perf_counter()
    rows.append((t2 - t_prev, t1 - t0, t2 - t1, dict(m.timings)))
    t_prev = t2
while res:
    res.pop(0).wait()
per = np.array([r[0] for r in rows]) * 1e3
print("period ms:", " ".join(f"{x:.1f}" for x in per))
print("run    ms:", " ".join(f"{r[1] * 1e3:.1f}" for r in rows))
print("wait   ms:", " ".join(f"{r[2] * 1e3:.1f}" for r in rows))
print("E wait ms:", " ".join(f"{r[3].get('host_wait_eigenvalues', 0) * 1e3:.1f}" for r in rows))
print(f"mean period of the last {n - 5}: {per[5:].mean():.2f} ms")
