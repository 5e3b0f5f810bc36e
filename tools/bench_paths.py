"""Secondary-path measurements (not the bench.py contract): IW reduction (C5), MMD statistic, prod-marginals,
D=64 pair kernel.  Prints one line per path with the achieved rate against its roofline."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvae import estimators as E, _native as N
from lvae.two_sample import MMDStatistic

def timed(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e-3

def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    # C5: IW reduction
    B, K, D = 16384, 1000, 32
    mu = torch.randn(B, D, device="cuda", generator=g); lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1
    eps = torch.randn(B, K, D, device="cuda", generator=g); lpxz = 5 * torch.randn(B, K, device="cuda", generator=g) - 100
    t = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=False))
    by = B * K * (4 * D + 4 + 8)      # eps + log_p_x_z read, log_q_z_x + log_p_z written
    print(f"IW C5 no-z : {t*1e3:.3f} ms  {B*K/t:.3e} (sentence,sample)/s  {by/t/1e9:.0f} GB/s algorithmic ({by/t/8e12*100:.1f}% of 8 TB/s)", flush=True)
    t = timed(lambda: E.reparam_iw(mu, lv, eps, lpxz, return_z=True))
    by = B * K * (8 * D + 4 + 8)
    print(f"IW C5 +z   : {t*1e3:.3f} ms  {B*K/t:.3e} /s  {by/t/1e9:.0f} GB/s ({by/t/8e12*100:.1f}% of 8 TB/s)", flush=True)
    del eps
    # MMD
    for n in (16384, 65536):
        x = torch.randn(n, 32, device="cuda", generator=g); y = torch.randn(n, 32, device="cuda", generator=g) * 0.9
        al = [0.1 * i for i in range(15)]
        t = timed(lambda: MMDStatistic(n, n)(x, y, al), iters=3, warm=1)
        ent = 3.0 * n * n
        # 3 n^2 nominal entries; the xx / yy sums walk the upper triangle only, so 2 n^2 are evaluated, at
        # 2 D flop each (the contraction is the cross term -2 x.y, the norms enter as biases)
        print(f"MMD tts n1=n2={n}: {t*1e3:.2f} ms  {ent/t:.3e} nominal kernel entries/s  {2.0*n*n*2*32/t/157.3e12*100:.1f}% of fp32 MFMA peak on the evaluated entries (2D flop each)", flush=True)
        t = timed(lambda: E.maximum_mean_discrepancy(y, x), iters=3, warm=1)
        print(f"MMD biased n={n}: {t*1e3:.2f} ms  {ent/t:.3e} entries/s", flush=True)
    # prod marginals + D=64
    for (n, d, pm, path) in ((16384, 32, True, 0), (65536, 64, False, 1), (65536, 16, False, 1), (16384, 128, False, 1)):
        mu = torch.randn(n, d, device="cuda", generator=g); lv = 0.5 * torch.randn(n, d, device="cuda", generator=g) - 1
        z = mu + torch.randn(n, d, device="cuda", generator=g) * torch.exp(0.5 * lv)
        t = timed(lambda: E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=pm, reduce_mean=False, path=path), iters=3, warm=1)
        print(f"log_qz N=M={n} D={d} pm={pm}: {t*1e3:.2f} ms  {n*n/t:.3e} pairs/s  {n*n*4*d/t/157.3e12*100:.1f}% of fp32 MFMA peak; {n*n*d/t:.3e} (i,j,d) terms/s", flush=True)

main()
