"""Second throughput scan: MI bounds, KDE, KS, permutation test, KL, encode_tail at odd shapes."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E, mutual_information as MI, kde as KD, pareto as PA, encoder_ops as EO

def timed(fn, reps=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
print("== representational MI bounds (K x K pairs)", flush=True)
for (K, D) in ((2000, 32), (2000, 20), (2000, 100), (2000, 768), (20000, 32), (20000, 64), (333, 32)):
    z = R(K, D); ml = torch.cat([R(K, D), 0.5 * R(K, D) - 1], 1)
    t = timed(lambda: MI.calc_upper_and_lower_bound_representational_mi(z, ml))
    print(f"K={K} D={D}: {t*1e3:9.3f} ms  {K*K*D/t:.3e} terms/s", flush=True)
print("== generative MI bounds (K x vocab)", flush=True)
for (K, V) in ((128, 50265), (512, 50265), (64, 1000)):
    probs = torch.softmax(R(K, 1, V), -1); samp = torch.randint(0, V, (K,), device="cuda", generator=g)
    t = timed(lambda: MI.calc_upper_and_lower_bound_generative_mi(probs, sampled_predictions=samp))
    print(f"K={K} V={V}: {t*1e3:9.3f} ms  {K*K/t:.3e} pairs/s  {K*V*4/t/1e9:.0f} GB/s of probs", flush=True)
print("== 1-D KDE marginal KL", flush=True)
for (n, D) in ((4096, 32), (42068, 32), (42068, 64), (4096, 768), (100000, 8)):
    lat = R(n, D); lp = -0.5 * (lat ** 2).sum(1)
    t = timed(lambda: E.get_1d_kde_dim_marg_kl_estimate(lat, lp), reps=2)
    print(f"n={n} D={D}: {t*1e3:9.3f} ms  {(n*D)**2/t:.3e} pairs/s", flush=True)
print("== KS two-sample", flush=True)
for n in (1000, 42068, 1 << 20, 1 << 24):
    a, b = R(n), R(n) + 0.01
    t = timed(lambda: PA.ks_2samp(a, b))
    print(f"n={n}: {t*1e3:9.3f} ms  {2*n/t:.3e} elements/s", flush=True)
print("== KL divergence", flush=True)
for (n, D) in ((32, 32), (42068, 32), (1 << 20, 64), (4096, 768)):
    mu, lv = R(n, D), 0.5 * R(n, D) - 1
    t = timed(lambda: E.kl_divergence(mu, lv, 0.5))
    print(f"n={n} D={D}: {t*1e3:9.3f} ms  {n*D*8/t/1e9:.0f} GB/s", flush=True)
print("== permutation test", flush=True)
for (n, D, P) in ((4096, 32, 100), (4096, 20, 100), (4096, 100, 100), (16384, 32, 1000), (16384, 256, 100), (1000, 32, 10000)):
    x, y = R(n, D), R(n, D) + 0.05
    labels = E.permutation_labels(n, n, P, "cuda", g)
    t = timed(lambda: E.mmd_permutation_test(x, y, [0.1 * i for i in range(15)], labels=labels), reps=2)
    print(f"n={n} D={D} P={P}: {t*1e3:9.3f} ms  {4*n*n*P/t:.3e} entry-labelings/s", flush=True)
print("== encode_tail forward + backward (general / default)", flush=True)
for (B, K, D) in ((64, 1, 32), (64, 100, 32), (64, 8, 64), (256, 10, 32), (32, 1, 768), (256, 1, 100), (2048, 1, 64)):
    mu = R(B, D).requires_grad_(True); lv = (0.5 * R(B, D) - 1).requires_grad_(True); eps = R(B, K, D)
    def step():
        mu.grad = lv.grad = None
        o = EO.encode_tail(mu, lv, n_samples=K, eps=eps)
        (o["log_q_z_x"].mean() - o["log_q_z"] + o["log_q_z_prod_marg"] - o["log_p_z"].mean() + o["mmd_loss"]).backward()
    t = timed(step, reps=10)
    print(f"B={B} K={K} D={D}: {t*1e3:9.3f} ms", flush=True)
