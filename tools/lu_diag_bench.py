"""Times tmf_diag_inverse_batched alone on 1024 synthetic always-blocks (k = 286, mb = mk = 312): HIP events around 5 launches
per outer step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from temfpy_amd import _native as nat  # noqa: E402




lib = nat.load()
n, k, mb = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 286, 312
el = 16
rng = np.random.default_rng(0)
W1 = (np.eye(mb) + 0.05 * (rng.normal(size=(mb, mb)) + 1j * rng.normal(size=(mb, mb)))).astype(np.complex128)
dW = torch.from_numpy(np.ascontiguousarray(W1.T)).cuda().repeat(n, 1, 1).contiguous()   # column-major copies
ddet = torch.zeros(n, dtype=torch.complex128, device="cuda")
dinv = torch.zeros(n * 64 * 64, dtype=torch.complex128, device="cuda")
stats = torch.zeros(2 * n, dtype=torch.float64, device="cuda")
ld = np.zeros(n, nat.diaginv_desc)
ld["W"] = dW.data_ptr() + np.arange(n) * mb * mb * el
ld["det"] = ddet.data_ptr() + np.arange(n) * el
ld["inv"] = dinv.data_ptr() + np.arange(n) * 64 * 64 * el
ld["mb"], ld["mk"], ld["k"], ld["ldw"] = mb, mb, k, mb
t = torch.from_numpy(ld.view(np.uint8)).cuda()
stream = torch.cuda.current_stream().cuda_stream
for step in range(-(-k // 64)):
    for _ in range(2):
        nat.check(lib.tmf_diag_inverse_batched(nat.TMF_C128, t.data_ptr(), n, step, stats.data_ptr(), stream), "diag_inverse")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        nat.check(lib.tmf_diag_inverse_batched(nat.TMF_C128, t.data_ptr(), n, step, stats.data_ptr(), stream), "diag_inverse")
    e1.record()
    torch.cuda.synchronize()
    print(f"step {step}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch", flush=True)
