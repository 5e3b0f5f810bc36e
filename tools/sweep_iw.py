"""C5 IW reduction under the launcher's development switches (LVAE_IW_U, LVAE_IW_PIPE), same process, same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or [(16384, 1000, 32)]
g = torch.Generator(device="cuda").manual_seed(0)
for B, K, D in shapes:
    mu = torch.randn(B, D, device="cuda", generator=g); lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1
    eps = torch.randn(B, K, D, device="cuda", generator=g); lpxz = 5 * torch.randn(B, K, device="cuda", generator=g) - 100
    for rnd in range(2):
        for u in ("8", "4", "2"):
            for pipe in ("1", "0"):
                os.environ["LVAE_IW_U"], os.environ["LVAE_IW_PIPE"] = u, pipe
                __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
                t = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False))
                t2 = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=True))
                print(f"B={B} K={K} D={D} U={u} pipe={pipe}: no-z {t:.3f} ms   +z {t2:.3f} ms", flush=True)
    del eps, lpxz
