"""Whole-call time of the C2 statistics step against the rows per workgroup of the operand preparation
(LVAE_PREP_RB 32-row blocks): sweep_prep_rb.py [n]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import distributed as LD
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(n, 32, device="cuda", generator=g); lv = 0.5 * torch.randn(n, 32, device="cuda", generator=g) - 1.0
z = mu + torch.randn(n, 32, device="cuda", generator=g) * torch.exp(0.5 * lv)
f = lambda: LD.hip_local_stats(mu, lv, z, mu, lv, math.log(n), False)
for _ in range(10): f()
for rb in (4, 1, 2, 8, 4):
    os.environ["LVAE_PREP_RB"] = str(rb)
    __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): s, _ = f()
    torch.cuda.synchronize()
    print(f"n={n} rb={rb}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms  KL {(s.tolist()[0]-s.tolist()[2])/n:.6f}", flush=True)
