"""H -> C at the benchmark size: host eigh (reference path) vs the GEMM-only sign iteration on the GPU."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from tests_inputs import random_hopping
from temfpy_amd import slater
H = random_hopping(1024, 0)
t0 = time.perf_counter(); C0, N0 = slater.correlation_matrix(H); t_host = time.perf_counter() - t0
eng = slater._engine("cuda:0")
eng.negative_projector(H)
t0 = time.perf_counter(); C1, steps = eng.negative_projector(H); t_dev = time.perf_counter() - t0
print(f"L=1024: host eigh {t_host*1e3:.0f} ms; device sign iteration {t_dev*1e3:.1f} ms ({steps} steps, incl. upload/download); "
      f"max |dC| = {np.abs(C1 - C0).max():.1e}, N = {N0}, tr C1 = {np.trace(C1).real:.6f}")
