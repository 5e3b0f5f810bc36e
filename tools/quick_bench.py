import sys, time, math
import numpy as np, torch
sys.path.insert(0, ".")
from lvae import estimators as E, _native as N
def run(Nn, M, D, path, pm=False, iters=5):
    g = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.randn(Nn, D, device="cuda", generator=g)
    logvar = 0.5 * torch.randn(Nn, D, device="cuda", generator=g) - 1.0
    eps = torch.randn(M, D, device="cuda", generator=g)
    idx = torch.arange(M, device="cuda") % Nn
    z = mu[idx] + eps * torch.exp(0.5 * logvar[idx])
    for _ in range(2):
        E.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=pm, reduce_mean=False, path=path)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); E.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=pm, reduce_mean=False, path=path); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    t = np.median(ts) * 1e-3
    pairs = Nn * M / t
    print(f"N={Nn} M={M} D={D} path={path} pm={pm}: {t*1e3:.3f} ms  {pairs:.3e} pairs/s  {pairs*4*D/157.3e12*100:.1f}% of fp32 MFMA peak", flush=True)
if __name__ == "__main__":
    run(4096, 4096, 32, 1)
    run(65536, 65536, 32, 1)
    run(65536, 65536, 32, 2)
    run(65536, 65536, 64, 1)
    run(16384, 16384, 32, 2, pm=True)
    run(65536, 65536, 32, 2, pm=True, iters=3)
