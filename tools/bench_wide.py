import math, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from lvae import estimators as E
for (n, D) in ((16384, 128), (16384, 256), (16384, 768), (8192, 768), (42068, 768)):
    g = torch.Generator(device="cuda").manual_seed(1)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    for _ in range(2): E.log_qz_raw(mu, lv, z, math.log(n), False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 3
    for _ in range(reps): E.log_qz_raw(mu, lv, z, math.log(n), False)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    print(f"n={n} D={D}: {t*1e3:.2f} ms  {n*n/t:.3e} pairs/s  {4*D*n*n/t/1e12:.1f} TFLOP/s-equivalent (4D flop per pair)", flush=True)
