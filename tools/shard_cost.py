"""Device time of one rank's share of the benchmark chain for world sizes 1, 2, 4, 8 (shards as bench.py --gpus N cuts them),
run one after the other on ONE GPU: what the per-rank fixed cost does to the strong-scaling curve."""
import os
import sys
import time

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

L, chi = 1024, 512
C, _ = slater.correlation_matrix(random_hopping(L, 0))
d_C = torch.from_numpy(np.ascontiguousarray(C).reshape(-1)).to("cuda:0")
tr = to_stopping_condition({"chi_max": chi})
eng = Engine("cuda:0")
for world in (1, 2, 4, 8):
    worst, rows = 0.0, []
    for rng in shard_sites(L, L // 2, world):
        for _ in range(2):
            eng.run(d_C, tr, L // 2, L, download=False, site_range=rng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m = eng.run(d_C, tr, L // 2, L, download=False, site_range=rng)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        worst = max(worst, dt)
        rows.append(f"{rng}: {dt:.1f}")
    print(f"world {world}: slowest shard {worst:.1f} ms -> {L / worst * 1e3:.0f} sites/s device-resident | " + "  ".join(rows), flush=True)
print("stages of the last shard:", {k: round(v * 1e3, 2) for k, v in m.timings.items()})
