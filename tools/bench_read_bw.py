"""What a pure streaming READ reaches on this GPU (the ceiling for the no-z IW reduction): torch.sum and
lvae_reduce_f32 over 2 GiB, and a device-to-device copy for comparison."""
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

x = torch.randn(16384 * 1000 * 32, device="cuda")
y = torch.empty_like(x)
nb = x.numel() * 4
t = timed(lambda: x.sum()); print(f"torch.sum      : {t*1e3:.3f} ms  {nb/t/1e9:.0f} GB/s read")
t = timed(lambda: N.reduce_f64(x, 0, 1.0)); print(f"lvae_reduce_f32: {t*1e3:.3f} ms  {nb/t/1e9:.0f} GB/s read")
t = timed(lambda: y.copy_(x)); print(f"copy           : {t*1e3:.3f} ms  {2*nb/t/1e9:.0f} GB/s read+write")
t = timed(lambda: x.abs_()); print(f"in-place abs   : {t*1e3:.3f} ms  {2*nb/t/1e9:.0f} GB/s read+write")
