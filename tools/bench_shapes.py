"""Per-rank shapes of the weak-scaling bench, timed on one GPU (kernel time through the profiling hooks)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import _native as N, distributed as LD
D = 32
for world in (1, 2, 4, 8):
    q = 256 * world
    n = max(q, int(round(65536 * math.sqrt(world) / q)) * q)
    m = n // world
    g = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu[:m] + torch.randn(m, D, device="cuda", generator=g) * torch.exp(0.5 * lv[:m])
    for _ in range(2):
        LD.hip_local_stats(mu, lv, z, mu[:m], lv[:m], math.log(n), False)
    torch.cuda.synchronize()
    N.check(N.lib().lvae_profile_begin(10))
    for _ in range(10):
        LD.hip_local_stats(mu, lv, z, mu[:m], lv[:m], math.log(n), False)
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
    N.check(N.lib().lvae_profile_end(ctypes.byref(ms), ctypes.byref(cnt)))
    t = ms.value / cnt.value * 1e-3
    print(f"world={world}: per-rank M={m} x N={n}: kernel {t*1e3:.3f} ms  {m*n*4*D/t/157.3e12*100:.1f}% of fp32 MFMA peak", flush=True)
