"""Per-site smallest pivot of the block-local LU (TMF_LU_DEBUG=1 prints every site below 1e-2 on stderr)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TMF_LU_DEBUG"] = "1"
import numpy as np
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.schmidt_utils import to_stopping_condition
from tests_inputs import random_hopping
e = Engine("cuda:0")
L = int(sys.argv[1]); chi = int(sys.argv[2])
C = slater.correlation_matrix(random_hopping(L, 0))[0]
e.run(C, to_stopping_condition({"chi_max": chi}), L // 2, L)
e._stage_timings()
print("min pivot", e.kernel_info.lu_min_pivot, "max |D^-1|", e.kernel_info.lu_max_inverse, "fallbacks", e.kernel_info.lu_fallbacks)
