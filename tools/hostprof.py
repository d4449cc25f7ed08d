import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from tests_inputs import random_hopping
from temfpy_amd import slater
from temfpy_amd.engine import Engine
from temfpy_amd.schmidt_utils import to_stopping_condition
L, chi = 1024, 512
C, N = slater.correlation_matrix(random_hopping(L, 0))
eng = Engine("cuda:0", profile=False)
d_C = torch.from_numpy(np.ascontiguousarray(C).reshape(-1)).to("cuda:0")
tr = to_stopping_condition({"chi_max": chi})
for _ in range(2):
    eng.run(d_C, tr, L // 2, L, download=False)
pr = cProfile.Profile()
pr.enable()
eng.run(d_C, tr, L // 2, L, download=False)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
