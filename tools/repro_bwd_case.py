"""Replays one case of tests/fuzz_bwd.py with all three routings: repro_bwd_case.py CASE"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_grad_mfma import _reference_grads, _our_grads
target = int(sys.argv[1])
rng = np.random.default_rng(2024)
for c in range(target + 1):
    M = int(rng.integers(1, 1800)); N = int(rng.integers(1, 1800)); D = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128]))
    sig = float(rng.choice([1.0, 0.5, 0.3])); mus = float(rng.choice([0.3, 1.0, 2.0]))
    far = rng.random() < 0.3; zero = rng.random() < 0.3
    if c < target: continue
    g = torch.Generator(device="cuda").manual_seed(c)
    mu = torch.randn(N, D, device="cuda", generator=g) * mus
    lv = 0.5 * torch.randn(N, D, device="cuda", generator=g) - 1.0 + 2 * math.log(sig)
    src = torch.randint(0, N, (M,), device="cuda", generator=g)
    z = mu[src] + torch.randn(M, D, device="cuda", generator=g) * torch.exp(0.5 * lv[src])
    if far: z[: max(1, M // 7)] += 6.0
    a = torch.randn(M, device="cuda", generator=g)
    if zero: a[::3] = 0.0
    print(f"case {c}: M={M} N={N} D={D} sigma x{sig} mu x{mus} far={far} zero={zero}")
    want = _reference_grads(mu, lv, z, a, math.log(N), chunk=64)
    for mode in ("mfma", "valu", None):
        if mode: os.environ["LVAE_BWD_PATH"] = mode; __import__("lvae._native", fromlist=["reload_env"]).reload_env()
        else: os.environ.pop("LVAE_BWD_PATH", None); __import__("lvae._native", fromlist=["reload_env"]).reload_env()
        got = _our_grads(mu, lv, z, a)
        errs = []
        for gg, w in zip(got, want):
            d = (gg.double() - w).abs()
            idx = int(d.argmax())
            errs.append(f"{float(d.max() / w.abs().max()):.2e}@row{idx // w.shape[1]}")
        print("  ", mode, errs)
