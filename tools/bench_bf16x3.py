"""Times the experimental split-bf16 pair path against the fp32 MFMA path (whole lvae_log_qz_f32 call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

def run(n, D, path, reps=20):
    g = torch.Generator(device="cuda").manual_seed(1)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    for _ in range(3):
        E.log_qz_raw(mu, lv, z, 0.0, False, path=path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        E.log_qz_raw(mu, lv, z, 0.0, False, path=path)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for n, D in ((65536, 32), (65536, 16), (16384, 32), (131072, 32), (65536, 64), (32768, 64)):
    t1, t3 = run(n, D, 1), run(n, D, 3)
    print(f"n={n} D={D}: fp32 MFMA {t1:.3f} ms ({n*n/t1*1e-9:.2f} Gpair/ms-> {n*n/t1/1e9:.1f}e12/s)  bf16x3 {t3:.3f} ms ({n*n/t3/1e9:.2f}e12 pairs/s)  x{t1/t3:.2f}", flush=True)

# kernel-sum / MMD entry points: forced fp32, forced split-bf16, and what LVAE_PATH_AUTO picks
from lvae import _native as N

def mmd_ms(n, D, unbiased, alphas, reps=10, path=N.PATH_MFMA):
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(n, D, device="cuda", generator=g)
    y = torch.randn(n, D, device="cuda", generator=g) * 1.02
    for _ in range(2):
        E.mmd_rbf(x, y, alphas, 1e-5, unbiased=unbiased, path=path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = E.mmd_rbf(x, y, alphas, 1e-5, unbiased=unbiased, path=path)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, float(out[0])

for n, D in ((65536, 32), (16384, 32), (8192, 32), (65536, 64)):
    for name, unb, alphas in (("tts (15 bandwidths)", True, [0.1 * i for i in range(15)]), ("biased (1 bandwidth)", False, [1.0 / D ** 2])):
        t1, v1 = mmd_ms(n, D, unb, alphas)
        t3, v3 = mmd_ms(n, D, unb, alphas, path=N.PATH_MFMA_BF16X3)
        ta, va = mmd_ms(n, D, unb, alphas, path=N.PATH_AUTO)
        print(f"MMD {name} n1=n2={n} D={D}: fp32 {t1:.2f} ms  bf16x3 {t3:.2f} ms  x{t1/t3:.2f}  AUTO (gated) {ta:.2f} ms   stat {v1:.6e} vs {v3:.6e} vs {va:.6e}", flush=True)
