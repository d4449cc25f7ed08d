"""cProfile of one warm pfaffian.C_to_MPS conversion at BASELINE config 4 (random BdG chain): where the host time of the
Python driver of the Pfaffian path goes."""
import cProfile
import io
import os
import pstats
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests", "golden"))
from make_golden_pfaffian import random_majorana_H  # noqa: E402
from temfpy_amd import pfaffian  # noqa: E402

L, chi = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 256
C = pfaffian.correlation_matrix(random_majorana_H(L, 0), "M->M")
for _ in range(2):
    pfaffian.C_to_MPS(C, {"chi_max": chi}, basis="M")
pr = cProfile.Profile()
pr.enable()
mps = pfaffian.C_to_MPS(C, {"chi_max": chi}, basis="M")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue())
