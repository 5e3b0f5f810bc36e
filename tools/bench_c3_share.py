"""One rank's share of BASELINE configs[2] (N = M = 1,048,576, D = 64 row-sharded over 8 GPUs): 131,072 local
samples against all 1,048,576 posteriors, timed on one GPU (the all-gather is not part of this)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import _native as N, distributed as LD
D, n, m = 64, 1048576, 131072
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(n, D, device="cuda", generator=g)
lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
z = mu[:m] + torch.randn(m, D, device="cuda", generator=g) * torch.exp(0.5 * lv[:m])
LD.hip_local_stats(mu, lv, z, mu[:m], lv[:m], math.log(n), False)
torch.cuda.synchronize()
N.check(N.lib().lvae_profile_begin(3))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(3):
    sums, _ = LD.hip_local_stats(mu, lv, z, mu[:m], lv[:m], math.log(n), False)
ev1.record()
torch.cuda.synchronize()
ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
N.check(N.lib().lvae_profile_end(ctypes.byref(ms), ctypes.byref(cnt)))
t = ms.value / cnt.value * 1e-3
tw = ev0.elapsed_time(ev1) / 3 * 1e-3
s = sums.tolist()
print(f"C3 per-rank share M={m} x N={n}, D={D}: kernel {t*1e3:.1f} ms ({m*n*4*D/t/157.3e12*100:.1f}% of fp32 MFMA peak), "
      f"whole call {tw*1e3:.1f} ms -> 8 GPUs: {8*m*n/tw:.3e} pair-evals/s if the all-gather (512 MB) hides or adds < 1 ms; "
      f"marginal KL of the share {(s[0]-s[2])/s[4]:.4f}", flush=True)
