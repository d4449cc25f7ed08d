"""Can a process that has initialised the GPU start a child process (fork + exec) on this pool?  Tests and the
worker pool rely on the answer (the pool itself is started before the parent touches a GPU either way)."""
import subprocess
import sys

import torch

x = torch.zeros(4, device="cuda:0") + 1
torch.cuda.synchronize()
r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True)
print("returncode", r.returncode, "stdout", r.stdout.strip(), "stderr", r.stderr.strip()[-300:])
