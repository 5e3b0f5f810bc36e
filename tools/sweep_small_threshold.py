import math, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from lvae import distributed as LD
for n in (4096, 5792, 6144, 8192, 12288, 16384):
    g = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.randn(n, 32, device="cuda", generator=g); lv = 0.5 * torch.randn(n, 32, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, 32, device="cuda", generator=g) * torch.exp(0.5 * lv)
    out = []
    for thr in ("0", "1e10"):
        os.environ["LVAE_SMALL_PAIRS"] = thr
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
        f = lambda: LD.hip_local_stats(mu, lv, z, mu, lv, math.log(n), False)
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): s, _ = f()
        torch.cuda.synchronize(); out.append(f"{'general' if thr == '0' else 'small'} {(time.perf_counter() - t0) / 50 * 1e6:.0f} us (KL {(s.tolist()[0]-s.tolist()[2])/n:.5f})")
    print(n, "; ".join(out), flush=True)
