"""1-D KDE at dataset scale (get_1d_kde_dim_marg_kl_estimate over all B*D latent coordinates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae.kde import GaussianKDE
for n in (32 * 1024, 256 * 1024, 42068 * 32):
    x = torch.randn(n, 1, device="cuda")
    k = GaussianKDE(X=x, bw="scott")
    for stab in (False, True):
        k.score_samples(x[:1024], stabilise=stab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k.score_samples(x, stabilise=stab)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"KDE 1-D n=m={n} stabilise={stab}: {dt*1e3:.1f} ms  {n*n/dt:.3e} pairs/s", flush=True)
