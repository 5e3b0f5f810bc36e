"""Throughput scan over odd shapes of the main entry points: looks for cliffs (a shape that runs far below its
neighbours).  Prints (i, j, d) terms/s for log q(z), pooled entries x D per second for MMD, GB/s for the IW reduction."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

def timed(fn, reps=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

g = torch.Generator(device="cuda").manual_seed(0)
def post(n, D):
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    return mu, lv

print("== log q(z): N posteriors, M samples", flush=True)
for (N, M, D) in ((16384, 16384, 1), (16384, 16384, 3), (16384, 16384, 5), (16384, 16384, 20), (16384, 16384, 48), (16384, 16384, 100),
                  (16384, 16384, 130), (128, 1 << 20, 32), (1 << 20, 128, 32), (33, 1 << 20, 32), (1 << 20, 33, 32), (1 << 22, 1024, 32),
                  (1000, 1000, 32), (3000, 3000, 64), (50000, 50000, 32), (100000, 7, 64), (7, 100000, 64)):
    mu, lv = post(N, D)
    z = torch.randn(M, D, device="cuda", generator=g)
    for pm in (False, True):
        if pm and N * M * D > 3e11: continue
        t = timed(lambda: E.log_qz_raw(mu, lv, z, math.log(N), pm))
        print(f"N={N} M={M} D={D} pm={int(pm)}: {t*1e3:9.3f} ms  {N*M/t:.3e} pairs/s  {N*M*D/t:.3e} terms/s", flush=True)
print("== MMD tts (15 bandwidths)", flush=True)
for (n, D) in ((16384, 1), (16384, 3), (16384, 20), (16384, 100), (16384, 300), (16384, 768), (100, 32), (1000, 32), (40000, 32), (1 << 17, 5)):
    x = torch.randn(n, D, device="cuda", generator=g); y = torch.randn(n, D, device="cuda", generator=g) + 0.1
    t = timed(lambda: E.mmd_rbf(x, y, [0.1 * i for i in range(15)], 1e-5, True))
    print(f"n={n} D={D}: {t*1e3:9.3f} ms  {4*n*n/t:.3e} pooled entries/s  {4*n*n*D/t:.3e} entry-dims/s", flush=True)
print("== IW reduction", flush=True)
for (B, K, D) in ((16384, 1000, 20), (16384, 1000, 48), (4096, 1000, 100), (200000, 10, 32), (1 << 20, 1, 32), (64, 100000, 32), (8, 1 << 20, 32), (16384, 1000, 36)):
    mu, lv = post(B, D)
    eps = torch.randn(B, K, D, device="cuda", generator=g); px = torch.randn(B, K, device="cuda", generator=g) - 100
    t = timed(lambda: E.reparam_iw(mu, lv, eps, px, return_z=False, return_densities=False))
    t2 = timed(lambda: E.reparam_iw(mu, lv, eps, px, return_z=True, return_densities=True))
    print(f"B={B} K={K} D={D}: reduction {t*1e3:8.3f} ms {B*K*(4*D+4)/t/1e9:6.0f} GB/s   everything {t2*1e3:8.3f} ms {B*K*(8*D+12)/t2/1e9:6.0f} GB/s", flush=True)
