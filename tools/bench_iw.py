"""IW reduction (C5: B = 16,384 sentences x K = 1,000 samples, D = 32) timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lvae import estimators as E

def timed(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e-3

# usage: python tools/bench_iw.py [B,K,D ...] [only=iw|dens|z]   (one form alone: for a kernel trace per form)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or \
    [(16384, 1000, 32), (16384, 600, 32), (4096, 1000, 64), (65536, 100, 32)]
only = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("only=")), None)
g = torch.Generator(device="cuda").manual_seed(0)
for B, K, D in shapes:
    mu = torch.randn(B, D, device="cuda", generator=g); lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1
    eps = torch.randn(B, K, D, device="cuda", generator=g); lpxz = 5 * torch.randn(B, K, device="cuda", generator=g) - 100
    if only:
        fn = {"iw": lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False, return_densities=False),
              "dens": lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False),
              "z": lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=True)}[only]
        by1 = B * K * {"iw": 4 * D + 4, "dens": 4 * D + 4 + 8, "z": 8 * D + 4 + 8}[only]
        t1 = timed(fn, reps=30)
        print(f"B={B} K={K} D={D} {only}: {t1*1e3:.3f} ms {by1/t1/1e9:.0f} GB/s ({by1/t1/8e12*100:.1f}%)", flush=True)
        del eps, lpxz
        continue
    t = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False))
    by = B * K * (4 * D + 4 + 8)      # eps + log_p_x_z read, log_q_z_x + log_p_z written
    t0 = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False, return_densities=False))
    by0 = B * K * (4 * D + 4)          # the reduction alone: eps + log_p_x_z read, iw_ll [B] written
    t2 = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=True))
    by2 = B * K * (8 * D + 4 + 8)
    print(f"B={B} K={K} D={D}: iw_ll only {t0*1e3:.3f} ms {by0/t0/1e9:.0f} GB/s ({by0/t0/8e12*100:.1f}%)   no-z {t*1e3:.3f} ms {by/t/1e9:.0f} GB/s ({by/t/8e12*100:.1f}%)   +z {t2*1e3:.3f} ms {by2/t2/1e9:.0f} GB/s ({by2/t2/8e12*100:.1f}%)", flush=True)
    del eps, lpxz
