"""Runs log_qz_raw a few times on one path at one size (for rocprofv3 passes): run_path.py PATH N D [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E
path, n, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator(device="cuda").manual_seed(1)
mu = torch.randn(n, D, device="cuda", generator=g)
lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
for _ in range(reps):
    E.log_qz_raw(mu, lv, z, 0.0, False, path=path)
torch.cuda.synchronize()
print("done", flush=True)
