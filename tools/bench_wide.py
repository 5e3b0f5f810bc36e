"""Wide latents (D > 128; the reference's module default is 768, modules/encoder.py:9): log q(z) with and without
prod-marginals, its backward at training-batch sizes, MMD and the IW reduction."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

def timed(fn, reps=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def inputs(n, D, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    return mu, lv, z

for (n, D) in ((16384, 128), (16384, 256), (16384, 768), (8192, 768), (42068, 768)):
    mu, lv, z = inputs(n, D)
    t = timed(lambda: E.log_qz_raw(mu, lv, z, math.log(n), False))
    print(f"log q(z) n={n} D={D}: {t*1e3:.2f} ms  {n*n/t:.3e} pairs/s  {n*n*D/t:.3e} (i,j,d) terms/s", flush=True)
for (n, D) in ((4096, 768), (16384, 256)):
    mu, lv, z = inputs(n, D)
    t = timed(lambda: E.log_qz_raw(mu, lv, z, math.log(n), True))
    print(f"log q(z) + prod-marginals n={n} D={D}: {t*1e3:.2f} ms  {n*n*D/t:.3e} (i,j,d) terms/s", flush=True)
for (B, D) in ((256, 768), (1024, 768), (4096, 768)):
    mu, lv, z = inputs(B, D)
    mu.requires_grad_(True); lv.requires_grad_(True)
    def step():
        mu.grad = lv.grad = None
        lq, pm = E.approximate_log_q_z(mu, lv, z, method="chen", dataset_size=42068, prod_marginals=True, reduce_mean=True)
        (lq + 0.5 * pm).backward()
    t = timed(step)
    print(f"log q(z) + prod-marginals, forward + backward B={B} D={D}: {t*1e3:.3f} ms", flush=True)
    x, y = z.detach(), torch.randn(B, D, device="cuda")
    t = timed(lambda: E.mmd_rbf(x, y, [0.1 * i for i in range(15)], 1e-5, True))
    print(f"MMD tts B={B} D={D}: {t*1e3:.3f} ms", flush=True)
B, K, D = 1024, 600, 768
mu, lv, _ = inputs(B, D)
eps = torch.randn(B, K, D, device="cuda"); px = torch.randn(B, K, device="cuda") - 100
t = timed(lambda: E.reparam_iw(mu, lv, eps, px, return_z=False, return_densities=False))
print(f"IW reduction B={B} K={K} D={D}: {t*1e3:.3f} ms  {B*K*(4*D+4)/t/1e9:.0f} GB/s", flush=True)
