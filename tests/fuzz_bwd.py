"""Randomised shape sweep of the backward of log q(z) (run by name on the GPU box: python tests/fuzz_bwd.py [cases]):
the matrix-core sweeps (forced) and the default routing against float64 autograd of the reference's formula.  Not
collected by pytest (no test_ prefix): minutes of GPU time."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from test_gpu_grad_mfma import _reference_grads, _our_grads

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(2024)
worst = 0.0
for c in range(cases):
    M = int(rng.integers(1, 1800)); N = int(rng.integers(1, 1800)); D = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128]))
    sig = float(rng.choice([1.0, 0.5, 0.3])); mus = float(rng.choice([0.3, 1.0, 2.0]))
    far = rng.random() < 0.3; zero = rng.random() < 0.3
    g = torch.Generator(device="cuda").manual_seed(c)
    mu = torch.randn(N, D, device="cuda", generator=g) * mus
    lv = 0.5 * torch.randn(N, D, device="cuda", generator=g) - 1.0 + 2 * math.log(sig)
    src = torch.randint(0, N, (M,), device="cuda", generator=g)
    z = mu[src] + torch.randn(M, D, device="cuda", generator=g) * torch.exp(0.5 * lv[src])
    if far:
        z[: max(1, M // 7)] += 6.0                      # far samples
    a = torch.randn(M, device="cuda", generator=g)
    if zero:
        a[::3] = 0.0
    want = _reference_grads(mu, lv, z, a, math.log(N), chunk=64)
    # float32 sweeps that sum the D terms of a pair's log-density directly (the vector-ALU form, like the reference's
    # own float32 autograd) carry a few ulp of |v| into every weight: far samples (|v| ~ 1e4) cost ~1e-3..1e-2 there
    # (allowance 2^-20 |v|max).  The matrix-core form renormalises the weights of a sample and keeps the partial sums
    # of its expansion small; a far sample still costs it one ulp of its own largest term (allowance 2^-24 |v|max).
    from lvae import estimators as E
    lq, _ = E.log_qz_raw(mu, lv, z, math.log(N), False)
    ulp_v = float(lq.abs().max()) * 2.0 ** -20
    for mode in ("mfma", None):
        if mode:
            os.environ["LVAE_BWD_PATH"] = mode
            __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
        else:
            os.environ.pop("LVAE_BWD_PATH", None)
            __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
        got = _our_grads(mu, lv, z, a)
        for name, gg, w in zip(("d_mu", "d_logvar", "d_z"), got, want):
            scale = max(float(w.abs().max()), 1e-30)
            err = float((gg.double() - w).abs().max()) / scale
            worst = max(worst, err)
            # d logvar = 0.5 sum_i W_ij ((z - mu)^2 / var - 1) cancels statistically when many samples come from the
            # posterior itself; the matrix-core form, which sums the three moments separately, is judged at 1e-3 there
            base = 1e-3 if (name == "d_logvar" and mode != "valu") else 2e-4
            if not (err <= base + (ulp_v / 16 if mode == "mfma" else ulp_v)) or not bool(torch.isfinite(gg).all()):
                print(f"FAIL case {c} mode {mode} M={M} N={N} D={D} sigma x{sig} {name}: rel err {err:.3e}", flush=True)
                sys.exit(1)
    if c % 10 == 9:
        print(f"{c + 1} cases clean, worst rel err {worst:.2e}", flush=True)
print(f"all {cases} cases clean, worst rel err {worst:.2e}")
