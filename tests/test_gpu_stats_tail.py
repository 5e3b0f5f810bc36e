"""The statistics call (lvae_latent_stats_f32) on the general path, whose last launch folds the redo merge, the
direct-form merge and the float64 sums (LateMerge): every source a row can take -- the fixed-reference result, the
redo pass after an underflow, the direct form behind the conditioning gate -- against the oracle and against the
launch-by-launch form (LVAE_NO_FUSED_TAIL=1)."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _case(kind, rng, N, M, D):
    mu = rng.standard_normal((N, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((N, D)) - 1.0).astype(np.float32)
    if kind == "peaked":                       # conditioning estimate far above the gate: direct form for every row
        lv = (0.3 * rng.standard_normal((N, D)) - 9.0).astype(np.float32)
    idx = rng.integers(0, N, M)
    z = (mu[idx] + rng.standard_normal((M, D)) * np.exp(0.5 * lv[idx])).astype(np.float32)
    if kind == "far":                          # some samples far from every posterior: their totals underflow the
        z[::7] += 9.0                          # fixed reference and are redone with the running maximum
    if kind == "spread":                       # a few extremely peaked posteriors among broad ones: the fixed-reference
        lv[:3] = -30.0                         # pass is switched off (lse_skip), every row comes from the redo pass
    return mu, lv, z, idx


@pytest.mark.parametrize("kind", ["plain", "far", "peaked", "spread"])
@pytest.mark.parametrize("D", [32, 16, 64])
def test_fused_tail_sources(kind, D, lvae_env):
    from lvae import distributed as LD
    rng = np.random.default_rng([D, len(kind), ord(kind[0])])      # (fixed per case; str hashes change per process)
    N, M = 700, 450
    mu, lv, z, idx = _case(kind, rng, N, M, D)
    mu_l, lv_l = mu[idx].copy(), lv[idx].copy()            # "own" posterior of every sample
    ref, _ = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False)
    lvae_env.set("LVAE_NO_SMALL_PATH", "1")          # the general path at a test-sized problem
    out = {}
    for fused in (True, False):
        if fused:
            lvae_env.unset("LVAE_NO_FUSED_TAIL")
        else:
            lvae_env.set("LVAE_NO_FUSED_TAIL", "1")
        sums, lq = LD.hip_local_stats(dev(mu), dev(lv), dev(z), dev(mu_l), dev(lv_l), math.log(N), False)
        out[fused] = (sums.cpu().numpy(), lq.cpu().numpy())
    s_f, lq_f = out[True]
    s_u, lq_u = out[False]
    assert np.isfinite(lq_f).all()
    np.testing.assert_array_equal(lq_f, lq_u)              # same arithmetic, launch for launch
    np.testing.assert_allclose(s_f, s_u, rtol=1e-12)
    scale = max(1.0, np.abs(ref).max())
    tol = 1e-4 * scale
    np.testing.assert_allclose(lq_f, ref, rtol=0, atol=tol)
    assert s_f[4] == M
    assert abs(s_f[0] - ref.astype(np.float64).sum()) <= 1e-4 * abs(ref.astype(np.float64).sum())
    lqx = onp.sample_log_likelihood(z, mu_l, lv_l, reduce_batch_dim=False).astype(np.float64).sum()
    lp = onp.sample_log_likelihood(z, reduce_batch_dim=False).astype(np.float64).sum()
    assert abs(s_f[3] - lqx) <= 1e-5 * abs(lqx) and abs(s_f[2] - lp) <= 1e-5 * abs(lp)


@pytest.mark.parametrize("kind", ["plain", "far", "peaked", "spread"])
@pytest.mark.parametrize("pm", [False, True])
def test_late_merge_of_the_plain_entry_point(kind, pm, lvae_env):
    """lvae_log_qz_f32 on the general path: the redo merge and the direct-form merge as ONE launch (the late merge
    alone), bit-equal to the two-launch form, with and without the product of marginals."""
    from lvae import estimators as E
    rng = np.random.default_rng([int(pm), len(kind), ord(kind[0])])
    N, M, D = 600, 500, 32
    mu, lv, z, _ = _case(kind, rng, N, M, D)
    ref, refpm = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    lvae_env.set("LVAE_NO_SMALL_PATH", "1")
    out = {}
    for fused in (True, False):
        if fused:
            lvae_env.unset("LVAE_NO_FUSED_TAIL")
        else:
            lvae_env.set("LVAE_NO_FUSED_TAIL", "1")
        lq, lqpm = E.approximate_log_q_z(dev(mu), dev(lv), dev(z), method="all", prod_marginals=pm, reduce_mean=False)
        out[fused] = (lq.cpu().numpy(), lqpm.cpu().numpy() if pm else None)
    np.testing.assert_array_equal(out[True][0], out[False][0])
    np.testing.assert_allclose(out[True][0], ref, rtol=0, atol=1e-4 * max(1.0, np.abs(ref).max()))
    if pm:
        np.testing.assert_array_equal(out[True][1], out[False][1])
        np.testing.assert_allclose(out[True][1], refpm, rtol=0, atol=1e-4 * max(1.0, np.abs(refpm).max()))
