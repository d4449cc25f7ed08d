"""Small driver for rocprofv3: N device-resident conversions of the benchmark workload (after warm-up)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping  # noqa: E402
from temfpy_amd import slater  # noqa: E402
from temfpy_amd.engine import Engine  # noqa: E402
from temfpy_amd.schmidt_utils import to_stopping_condition  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
C, _ = slater.correlation_matrix(random_hopping(L, 0))
eng = Engine("cuda:0")
tr = to_stopping_condition({"chi_max": chi})
for _ in range(n):
    eng.run(C, tr, L // 2, L, download=False)
print("done")
