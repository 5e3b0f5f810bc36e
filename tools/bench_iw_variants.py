"""lvae_reparam_iw_f32 at C5 with different sets of outputs / inputs, to see what the stream pays for."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lvae import _native as N

def timed(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e-3

B, K, D = 16384, 1000, 32
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(B, D, device="cuda", generator=g); lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1
eps = torch.randn(B, K, D, device="cuda", generator=g); lpxz = 5 * torch.randn(B, K, device="cuda", generator=g) - 100
lq = torch.empty(B, K, device="cuda"); lp = torch.empty(B, K, device="cuda"); iw = torch.empty(B, device="cuda")
z = torch.empty(B, K, D, device="cuda")
lib = N.lib(); st = N.stream_ptr(eps.device)
def call(px, zz, a, b, c):
    N.check(lib.lvae_reparam_iw_f32(N.ptr(mu), N.ptr(lv), N.ptr(eps), N.ptr(px), B, K, D, N.ptr(zz), N.ptr(a), N.ptr(b), N.ptr(c), st))
rd = B * K * D * 4
for name, args, extra in (("iw_ll only, no log_p_x_z", (None, None, None, None, iw), 0),
                          ("iw_ll only", (lpxz, None, None, None, iw), 4),
                          ("log_q_z_x + log_p_z + iw_ll", (lpxz, None, lq, lp, iw), 12),
                          ("log_q_z_x + log_p_z only", (None, None, lq, lp, None), 8),
                          ("z only", (None, z, None, None, None), 4 * D),
                          ("everything", (lpxz, z, lq, lp, iw), 12 + 4 * D)):
    t = timed(lambda: call(*args))
    by = rd + B * K * extra
    print(f"{name:32s}: {t*1e3:.3f} ms  {by/t/1e9:.0f} GB/s", flush=True)
