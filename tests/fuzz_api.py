"""Randomised shapes through every host-mirror entry point that sizes a workspace or picks a kernel by shape: no call may
be refused (workspace queries must cover what the call plans) and every result must be finite; where the NumPy oracle
is cheap the value is checked too.  Run by name on the GPU box -- `python tests/fuzz_api.py <cases> <seed>` -- not
collected by pytest."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lvae import estimators as E, mutual_information as MI, pareto as PA, kde as KD
from oracle import oracle_np as onp

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
bad = 0


def check(name, tag, fn, ref=None, tol=2e-4):
    global bad
    try:
        out = fn()
    except Exception as e:       # noqa: BLE001
        print(f"FAIL {name} raised {e!r}  {tag}", flush=True)
        bad += 1
        return
    vals = [float(v) for v in (out if isinstance(out, (tuple, list)) else [out])]
    if not all(math.isfinite(v) for v in vals):
        print(f"FAIL {name} non-finite {vals}  {tag}", flush=True)
        bad += 1
    elif ref is not None:
        want = [float(v) for v in (ref() if callable(ref) else ref)]
        for a, b in zip(vals, want):
            if abs(a - b) > tol * max(1.0, abs(b)):
                print(f"FAIL {name} {a} vs {b}  {tag}", flush=True)
                bad += 1


DS = [1, 2, 3, 5, 8, 16, 20, 31, 32, 33, 48, 64, 65, 100, 128, 130, 200, 300]
for c in range(cases):
    D = int(rng.choice(DS))
    K = int(rng.integers(2, 3000))
    tag = f"case {c}: K={K} D={D}"
    z, mu, lv = R(K, D), R(K, D), 0.5 * R(K, D) - 1.0
    small = K * K * D <= 4e7
    zn, mn, ln = (t.cpu().numpy() for t in (z, mu, lv))
    # MI bounds (two log q(z) plans in one region)
    check("mi_bounds", tag, lambda: MI.calc_upper_and_lower_bound_representational_mi(z, torch.cat([mu, lv], 1)),
          (lambda: onp.calc_upper_and_lower_bound_representational_mi(zn, np.concatenate([mn, ln], 1))) if small else None)
    # q_z_estimate (roles swapped) with a different sample count
    S = int(rng.integers(1, 2000))
    zs = R(S, D)
    check("q_z_estimate", tag + f" S={S}", lambda: MI.q_z_estimate(zs, mu, lv.exp()).double().mean(),
          (lambda: [onp.q_z_estimate(zs.cpu().numpy(), mn, np.exp(ln)).mean()]) if S * K * D <= 4e7 else None)
    # marginal KL / TC on ragged M != N
    M = int(rng.integers(1, 3000))
    zm = R(M, D)
    check("log_qz+pm", tag + f" M={M}", lambda: [t.double().mean() for t in E.approximate_log_q_z(mu, lv, zm, method="all", prod_marginals=True, reduce_mean=False)],
          (lambda: [a.mean() for a in onp.approximate_log_q_z(mn, ln, zm.cpu().numpy(), method="all", prod_marginals=True, reduce_mean=False)]) if M * K * D <= 4e7 else None)
    # MMD statistic and permutation test, n1 != n2
    n1, n2 = int(rng.integers(2, 3000)), int(rng.integers(2, 3000))
    Dm = min(D, 256)
    x, y = R(n1, Dm), R(n2, Dm) + 0.1
    alphas = [0.1 * i for i in range(15)]
    check("mmd_tts", tag + f" n1={n1} n2={n2}", lambda: E.mmd_rbf(x, y, alphas, 1e-5, True)[0],
          (lambda: [onp.mmd_statistic_tts(x.cpu().numpy(), y.cpu().numpy(), alphas)]) if n1 * n2 * Dm <= 2e7 else None, tol=1e-3)
    P = int(rng.integers(1, 40))
    check("mmd_perm", tag + f" n1={n1} n2={n2} P={P}", lambda: [E.mmd_permutation_test(x, y, alphas, n_permutations=P, generator=g)[k] for k in ("statistic", "pval")])
    # KDE and KS
    n = int(rng.integers(2, 400))
    lat = R(n, int(rng.choice([1, 3, 8, 32])))
    check("kde_kl", tag + f" n={n}", lambda: E.get_1d_kde_dim_marg_kl_estimate(lat, -0.5 * (lat ** 2).sum(1)).double().mean())
    a, b = R(int(rng.integers(1, 100000))), R(int(rng.integers(1, 100000)))
    check("ks", tag, lambda: PA.ks_2samp(a, b)[0])
    if (c + 1) % 20 == 0:
        print(f"{c + 1} cases, {bad} failures", flush=True)
print(f"{cases} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
