"""rocprofv3 driver: N device-resident conversions of ONE rank's share of the benchmark chain (world size and rank from the
command line): which kernels carry the per-rank fixed cost that caps strong scaling.
usage: python3 tools/profile_shard.py <world> <rank> [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_inputs import random_hopping  # noqa: E402
from temfpy_amd import slater  # noqa: E402
from temfpy_amd.engine import Engine  # noqa: E402
from temfpy_amd.multi_gpu import shard_sites  # noqa: E402
from temfpy_amd.schmidt_utils import to_stopping_condition  # noqa: E402

world, rank = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L, chi = 1024, 512
C, _ = slater.correlation_matrix(random_hopping(L, 0))
d_C = torch.from_numpy(np.ascontiguousarray(C).reshape(-1)).to("cuda:0")
tr = to_stopping_condition({"chi_max": chi})
eng = Engine("cuda:0")
rng = shard_sites(L, L // 2, world)[rank]
for _ in range(reps):
    m = eng.run(d_C, tr, L // 2, L, download=False, site_range=rng)
torch.cuda.synchronize()
print("range", rng, {k: round(v * 1e3, 2) for k, v in m.timings.items()})
