"""rocprofv3 driver: the Gutzwiller projection of BASELINE config 5 (L=512 uniform chain, spinful PH, chi 512)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import uniform_chain
from temfpy_amd import slater, gutzwiller
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
C, N = slater.correlation_matrix(uniform_chain(L))
mps = slater.C_to_MPS(C, {"chi_max": 512}, as_tenpy=False, spinful="PH")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    res = gutzwiller.abrikosov_ph(mps)
print({k: round(v * 1e3, 1) for k, v in res.timings.items()})
