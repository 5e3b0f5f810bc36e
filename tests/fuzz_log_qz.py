"""Randomised shape sweep of lvae_log_qz_f32 (all paths, with / without prod-marginals and diagonal exclusion)
and the kernel sums against the NumPy oracle.  Test infrastructure (it imports the oracle, so it lives under
tests/), run by name on the GPU box -- `python tests/fuzz_log_qz.py <seed> <cases> [big]` -- and not collected by
pytest; prints every failing case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lvae import _native as N
from lvae import estimators as E
from oracle import oracle_np as onp

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
big = len(sys.argv) > 3 and sys.argv[3] == "big"   # >= 2^22 pairs: the AUTO path's matrix-pipe prod-marginals sweep
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for c in range(cases):
    D = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 130]))
    Nn = int(rng.integers(1, 700))
    M = int(rng.integers(1, 700))
    if big:
        D = int(rng.choice([3, 16, 17, 32, 33, 48, 64, 80, 100, 128]))
        Nn, M = int(rng.integers(2048, 2700)), int(rng.integers(2048, 2700))
    mu = rng.standard_normal((Nn, D)).astype(np.float32) * rng.choice([0.1, 1.0, 3.0])
    lv = (rng.standard_normal((Nn, D)) * rng.choice([0.1, 0.5, 1.5]) + rng.choice([-3.0, -1.0, 0.0])).astype(np.float32)
    idx = rng.integers(0, Nn, M)
    z = (mu[idx] + rng.standard_normal((M, D)) * np.exp(0.5 * lv[idx]) * rng.choice([1.0, 3.0])).astype(np.float32)
    pm = bool(rng.integers(0, 2)) or big
    mus = mu.astype(np.float64) - mu.astype(np.float64).mean(0)
    cond = (0.5 * (2 * np.abs(mus) / np.exp(0.5 * lv.astype(np.float64)) + 3) ** 2).sum(1).max()   # the gate's estimate
    ref, refpm = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    for path in (0, 1, 2, 3):
        if path == 1 and D > 128: continue
        if path == 3 and D > 64: continue
        try:
            lq, lqpm = E.approximate_log_q_z(dev(mu), dev(lv), dev(z), method="all", prod_marginals=pm,
                                             reduce_mean=False, path=path)
            got = lq.cpu().numpy()
            scale = max(1.0, np.abs(ref).max())
            tol = 1e-4 * scale if path in (0, 2) else 3e-3 * scale     # forced MFMA paths: conditioning is the caller's risk
            if path == 0 and cond <= 2e4:      # AUTO below its gate: the matrix-core form, one ulp of the estimate
                tol = max(tol, 2.0 ** -23 * cond)
            if path in (1, 3):                 # forced, ungated: the same error model at whatever the estimate is
                tol = max(tol, 2.0 ** -23 * cond)    # (seed 1001 case 244: estimate 3.6e6, error 0.31 = 0.7 ulp of it)
            err = np.abs(got - ref).max()
            ok = np.isfinite(got).all() and err <= tol
            if pm:
                errpm = np.abs(lqpm.cpu().numpy() - refpm).max()
                # (path 3 runs the prod-marginals expansion ungated: conditioning is the caller's risk there too)
                ok = ok and errpm <= (3e-3 if path == 3 else 1e-4) * max(1.0, np.abs(refpm).max())
            if not ok:
                bad += 1
                print(f"FAIL case {c}: N={Nn} M={M} D={D} path={path} pm={pm} cond={cond:.3g} err={err:.3e} tol={tol:.1e}"
                      + (f" errpm={errpm:.3e} (scale {np.abs(refpm).max():.1f})" if pm else ""), flush=True)
        except N.LvaeError as e:
            bad += 1
            print(f"ERROR case {c}: N={Nn} M={M} D={D} path={path} pm={pm}: {e}", flush=True)
    # kernel sums (both math modes) on the same points
    if D <= 128:
        al = [0.1 * i for i in range(15)] if c % 2 else [0.3]
        a64, b64 = z.astype(np.float64), mu.astype(np.float64)
        d2 = 1e-5 + np.abs((a64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None] - 2 * a64 @ b64.T)
        want = sum(np.exp(-x * d2) for x in al).sum()
        for mode in (N.PATH_MFMA, N.PATH_MFMA_BF16X3):
            got = float(E.kernel_sum(dev(z), dev(mu), al, 1e-5, False, 0, path=mode))
            # (d2 comes from the fp32 expansion |a|^2 + |b|^2 - 2ab: its rounding, times alpha, is the floor)
            if not abs(got - want) <= 1e-4 * want + 1e-12:
                bad += 1
                print(f"FAIL ksum case {c}: na={M} nb={Nn} D={D} mode={mode} got={got} want={want}", flush=True)
print(f"{cases} cases, {bad} failures", flush=True)
