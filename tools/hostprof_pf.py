import cProfile, pstats, sys, os, io
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
from make_golden_pfaffian import kitaev_majorana_H
from temfpy_amd import pfaffian, testing
H = kitaev_majorana_H(512, 1.5j, 1j)
C = pfaffian.correlation_matrix(H, "M->M")
for _ in range(2):
    pfaffian.C_to_MPS(C, {"chi_max": 256}, basis="M")
pr = cProfile.Profile(); pr.enable()
pfaffian.C_to_MPS(C, {"chi_max": 256}, basis="M")
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
