"""Times the C2 MFMA pair kernel with the library named by LVAE_LIB (ablation builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lvae import estimators as E
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 65536
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(n, D, device="cuda", generator=g)
logvar = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * logvar)
for _ in range(3):
    E.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False, path=1)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); E.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False, path=1); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
t = np.median(ts) * 1e-3
print(f"{os.environ.get('LVAE_LIB','default'):50s} D={D}: {t*1e3:.3f} ms  {n*n*4*D/t/157.3e12*100:.1f}% of peak (whole call)", flush=True)
