"""How large the AUTO path's error gets just below the conditioning gate: random posteriors whose conditioning
estimate is steered to a target by scaling mu, log q(z) against the float64 NumPy oracle.
Test infrastructure (imports the oracle), run by name on the GPU box: `python tests/fuzz_gate.py <seed> <cases>`;
prints, per target, the worst error in nats and relative to max(1, |log q|max)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lvae import estimators as E
from oracle import oracle_np as onp

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cond_of(mu, lv):
    mus = mu.astype(np.float64) - mu.astype(np.float64).mean(0)
    return (0.5 * (2 * np.abs(mus) / np.exp(0.5 * lv.astype(np.float64)) + 3) ** 2).sum(1).max()


for target in (2.5e3, 5e3, 1e4, 1.9e4):
    worst_abs = worst_rel = 0.0
    for c in range(cases):
        D = int(rng.choice([2, 5, 8, 16, 17, 32, 33, 64, 100, 128]))
        Nn, M = int(rng.integers(32, 900)), int(rng.integers(32, 900))
        lv = (rng.standard_normal((Nn, D)) * rng.choice([0.1, 0.5]) + rng.choice([-5.0, -3.0, -1.0])).astype(np.float32)
        mu0 = rng.standard_normal((Nn, D))
        lo, hi = 0.0, 1e3
        for _ in range(60):                      # scale of mu that puts the estimate at the target
            mid = 0.5 * (lo + hi)
            if cond_of((mu0 * mid).astype(np.float32), lv) > target: hi = mid
            else: lo = mid
        mu = (mu0 * lo).astype(np.float32)
        if cond_of(mu, lv) > target * 1.001 or lo == 0.0:
            continue
        idx = rng.integers(0, Nn, M)
        z = (mu[idx] + rng.standard_normal((M, D)) * np.exp(0.5 * lv[idx]) * rng.choice([1.0, 2.0])).astype(np.float32)
        ref, _ = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False)
        lq, _ = E.approximate_log_q_z(dev(mu), dev(lv), dev(z), method="all", prod_marginals=False, reduce_mean=False)
        forced, _ = E.approximate_log_q_z(dev(mu), dev(lv), dev(z), method="all", prod_marginals=False, reduce_mean=False, path=2)
        got = lq.cpu().numpy()
        on_mfma = not np.array_equal(got, forced.cpu().numpy())
        err = np.abs(got - ref).max()
        rel = err / max(1.0, np.abs(ref).max())
        merr = abs(got.mean() - ref.mean()) / max(1.0, abs(ref.mean()))
        if err > worst_abs: worst_abs = err
        if rel > worst_rel:
            worst_rel = rel
            print(f"  target {target:.0f}: N={Nn} M={M} D={D} cond={cond_of(mu, lv):.0f} mfma={on_mfma} err={err:.2e} nats rel={rel:.2e} mean-rel={merr:.1e}", flush=True)
    print(f"target cond {target:.0f}: worst {worst_abs:.2e} nats, {worst_rel:.2e} relative", flush=True)
