import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lvae import estimators as E
def timed(fn, reps=21):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
g = torch.Generator(device="cuda").manual_seed(0)
for B, K, D in ((65536, 100, 32), (32768, 200, 32), (16384, 400, 32)):
    mu = torch.randn(B, D, device="cuda", generator=g); lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1
    eps = torch.randn(B, K, D, device="cuda", generator=g); lpxz = 5 * torch.randn(B, K, device="cuda", generator=g) - 100
    for tpb in ("256", "128", "64"):
        for u in ("8", "4", "2"):
            for pipe in ("1", "0"):
                os.environ["LVAE_IW_TPB"], os.environ["LVAE_IW_U"], os.environ["LVAE_IW_PIPE"] = tpb, u, pipe
                __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
                t0 = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False, return_densities=False))
                t = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False))
                t2 = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=True))
                print(f"B={B} K={K} tpb={tpb} U={u} pipe={pipe}: iw-only {t0:.3f}  no-z {t:.3f}  +z {t2:.3f} ms", flush=True)
    del eps, lpxz
