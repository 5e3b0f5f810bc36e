"""Randomised shape sweep of lvae_reparam_iw_f32 (every combination of optional inputs / outputs: log p(x|z) given or
not, z written or not, the two density matrices written or not, posterior given or the prior) against the NumPy oracle
in float64.  Test infrastructure (it imports the oracle, so it lives under tests/), run by name on the GPU box --
`python tests/fuzz_iw.py <cases> <seed>` -- and not collected by pytest; prints every failing case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lvae import _native as N
from oracle import oracle_np as onp

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
lib = N.lib()
bad = 0
worst = 0.0
for c in range(cases):
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 20, 31, 32, 33, 48, 63, 64, 65, 100, 128, 130, 256, 257, 300, 768, 1000, 1024, 1030]))
    K = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 250, 255, 256, 257, 600, 1000, 1031]))
    B = int(rng.integers(1, max(2, min(400, 2_000_000 // (K * D)))))
    mu = (rng.standard_normal((B, D)) * rng.choice([0.1, 1.0, 3.0])).astype(np.float32)
    lv = (rng.standard_normal((B, D)) * rng.choice([0.1, 0.5, 1.5]) + rng.choice([-3.0, -1.0, 0.0])).astype(np.float32)
    eps = rng.standard_normal((B, K, D)).astype(np.float32)
    lpxz = (5.0 * rng.standard_normal((B, K)) - 100.0 * rng.choice([0.1, 1.0, 10.0])).astype(np.float32)
    have_px, want_z, want_d, want_iw = (bool(rng.integers(0, 2)) for _ in range(4))
    if not (want_z or want_d or want_iw):
        want_iw = True
    t = {k: dev(v) for k, v in dict(mu=mu, lv=lv, eps=eps, px=lpxz if have_px else None).items()}
    z = torch.full((B, K, D), float("nan"), device="cuda") if want_z else None
    lq = torch.full((B, K), float("nan"), device="cuda") if want_d else None
    lp = torch.full((B, K), float("nan"), device="cuda") if want_d else None
    iw = torch.full((B,), float("nan"), device="cuda") if want_iw else None
    rc = lib.lvae_reparam_iw_f32(N.ptr(t["mu"]), N.ptr(t["lv"]), N.ptr(t["eps"]), N.ptr(t["px"]), B, K, D, N.ptr(z), N.ptr(lq),
                                 N.ptr(lp), N.ptr(iw), N.stream_ptr(t["eps"].device))
    torch.cuda.synchronize()
    tag = f"case {c}: B={B} K={K} D={D} px={have_px} z={want_z} dens={want_d} iw={want_iw}"
    if rc != 0:
        print("FAIL rc", rc, lib.lvae_last_error(), tag, flush=True)
        bad += 1
        continue
    m64, l64, e64 = mu.astype(np.float64), lv.astype(np.float64), eps.astype(np.float64)
    # like the reference (modules/encoder.py:148-157, then sample_log_likelihood on z): z is rounded to float32 first
    # and log q(z|x) is formed from (z - mu)^2 / var, not from eps^2
    z_ref = onp.reparameterize(mu, lv, eps).astype(np.float64)
    lq_ref = (-0.5 * (l64[:, None, :] + np.log(2 * np.pi) + (z_ref - m64[:, None, :]) ** 2 * np.exp(-l64)[:, None, :])).sum(-1)
    lp_ref = (-0.5 * (np.log(2 * np.pi) + z_ref ** 2)).sum(-1)
    lw = lp_ref - lq_ref + (lpxz.astype(np.float64) if have_px else 0.0)
    mx = lw.max(1, keepdims=True)
    iw_ref = (mx + np.log(np.exp(lw - mx).sum(1, keepdims=True))).ravel() - np.log(K)
    errs = []
    if want_z:
        errs.append(("z", np.abs(z.cpu().numpy() - z_ref).max() / max(1.0, np.abs(z_ref).max())))
    if want_d:
        # One ulp of z (its exponential may differ by an ulp from NumPy's) moves (z - mu)^2 / var by 2 |eps| ulp(z) / std
        # per dimension -- the reference's own formula has this sensitivity when std << |mu|; it is part of the bound.
        sens = (np.abs(e64) * 1.2e-7 * np.abs(z_ref) * np.exp(-0.5 * l64)[:, None, :]).sum(-1)
        errs.append(("log_q_z_x", (np.abs(lq.cpu().numpy() - lq_ref) / (np.maximum(1.0, np.abs(lq_ref).max()) + sens * 1e6)).max()))
        errs.append(("log_p_z", np.abs(lp.cpu().numpy() - lp_ref).max() / max(1.0, np.abs(lp_ref).max())))
    if want_iw:
        # (the log-weight is a difference of float32 sums of the size of the densities: that size is the error scale)
        scale = max(1.0, np.abs(lq_ref).max() + np.abs(lp_ref).max() + (np.abs(lpxz).max() if have_px else 0.0))
        scale += 1e6 * (np.abs(e64) * 1.2e-7 * np.abs(z_ref) * np.exp(-0.5 * l64)[:, None, :]).sum(-1).max()
        errs.append(("iw_ll", np.abs(iw.cpu().numpy() - iw_ref).max() / scale))
    for name, e in errs:
        worst = max(worst, float(e)) if np.isfinite(e) else worst
        # (D > 1024 takes the one-lane-per-row kernel: a serial float32 sum over the row, ~2e-6)
        if not np.isfinite(e) or e > (4e-6 if D > 1024 else 1e-6):
            print(f"FAIL {name} rel err {e:.3e}  {tag}", flush=True)
            bad += 1
    if (c + 1) % 50 == 0:
        print(f"{c + 1} cases, {bad} failures, worst rel err {worst:.2e}", flush=True)
print(f"{cases} cases, {bad} failures, worst rel err {worst:.2e}", flush=True)
sys.exit(1 if bad else 0)
