"""cProfile of one warm gutzwiller.abrikosov_ph at BASELINE config 5 (L = 512 uniform chain, spinful PH, chi = 512): where the
host time of the Gutzwiller projection goes."""
import cProfile
import io
import os
import pstats
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import uniform_chain  # noqa: E402
from temfpy_amd import slater, gutzwiller  # noqa: E402

L, chi = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 512
C, _ = slater.correlation_matrix(uniform_chain(L))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mps = slater.C_to_MPS(C, {"chi_max": chi}, as_tenpy=False, spinful="PH")
    for _ in range(2):
        gutzwiller.abrikosov_ph(mps)
    pr = cProfile.Profile()
    pr.enable()
    res = gutzwiller.abrikosov_ph(mps)
    pr.disable()
print({k: round(v * 1e3, 1) for k, v in res.timings.items()})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue())
