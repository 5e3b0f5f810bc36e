"""log q(z) (joint, forced fp32 MFMA path) across latent widths at one size: bench_dims.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E, _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for D in (16, 32, 48, 64, 96, 128):
    g = torch.Generator(device="cuda").manual_seed(1)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    for _ in range(3): E.log_qz_raw(mu, lv, z, 0.0, False, path=N.PATH_MFMA)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): E.log_qz_raw(mu, lv, z, 0.0, False, path=N.PATH_MFMA)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    DP = 16 if D <= 16 else 32 if D <= 32 else 64 if D <= 64 else 128
    print(f"n={n} D={D}: {t * 1e3:.2f} ms  {n * n / t:.3e} pairs/s  {100 * 4.0 * DP * n * n / t / 157.3e12:.1f}% of fp32 MFMA peak on the padded width ({100 * 4.0 * D * n * n / t / 157.3e12:.1f}% on D)", flush=True)
