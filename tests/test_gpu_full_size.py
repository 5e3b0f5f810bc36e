"""Oracle parity at BASELINE's FULL sizes.  The HIP path runs the whole configuration; the CPU oracle (plain C
with OpenMP, float64 NumPy) re-evaluates a random subset of its outputs -- every output of these estimators
depends on one row / column / sentence of the job only, so a subset is checked against exactly the value the
oracle would give inside the full run.  Each oracle leg is a few seconds of CPU.  Bar: 1e-4 relative on the
means (north star), element-wise tolerances stated per test."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle_c
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _posteriors(g, n, D):
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    return mu, lv


def test_c2_log_qz_rows_vs_oracle():
    """configs[1]: N = M = 65,536, D = 32, joint density and prod-marginals; 512 random sample rows against
    oracle_c.log_qz over ALL 65,536 posteriors (loss_and_optimisation.py:139-221)."""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(11)
    n, D = 65536, 32
    mu, lv = _posteriors(g, n, D)
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    lq, lqpm = E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    rows = torch.randperm(n, device="cuda", generator=g)[:512]
    want, want_pm = oracle_c.log_qz(z[rows].cpu().numpy(), mu.cpu().numpy(), lv.cpu().numpy(), math.log(n), True)
    got, got_pm = lq[rows].double().cpu().numpy(), lqpm[rows].double().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)
    np.testing.assert_allclose(got_pm, want_pm, rtol=0, atol=2e-3)
    assert abs(got.mean() - want.mean()) <= 1e-4 * abs(want.mean())
    assert abs(got_pm.mean() - want_pm.mean()) <= 1e-4 * abs(want_pm.mean())
    # the "chen" normaliser is the same vector shifted by log(dataset_size) (:207 vs :215)
    lq_chen, _ = E.approximate_log_q_z(mu, lv, z, method="chen", dataset_size=42068, prod_marginals=False,
                                       reduce_mean=False)
    assert float((lq_chen[rows] - lq[rows] + math.log(42068)).abs().max()) <= 1e-4


def test_c3_rank_share_vs_oracle():
    """configs[2] as ONE of its eight ranks sees it: 131,072 local samples against all N = 1,048,576 posteriors,
    D = 64 (the all-gather is covered by the gloo tests); 48 sample rows against oracle_c.log_qz."""
    from lvae import distributed as LD
    g = torch.Generator(device="cuda").manual_seed(12)
    n, m, D = 1048576, 131072, 64
    mu, lv = _posteriors(g, n, D)
    lo = 3 * m                                         # the share of rank 3
    z = mu[lo:lo + m] + torch.randn(m, D, device="cuda", generator=g) * torch.exp(0.5 * lv[lo:lo + m])
    sums, lq = LD.hip_local_stats(mu, lv, z, mu[lo:lo + m].contiguous(), lv[lo:lo + m].contiguous(), math.log(n), False)
    rows = torch.randperm(m, device="cuda", generator=g)[:48]
    want, _ = oracle_c.log_qz(z[rows].cpu().numpy(), mu.cpu().numpy(), lv.cpu().numpy(), math.log(n), False)
    got = lq[rows].double().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)
    assert abs(got.mean() - want.mean()) <= 1e-4 * abs(want.mean())
    s = sums.tolist()
    assert s[4] == m and abs(s[0] / m - float(lq.double().mean())) <= 1e-9 * abs(s[0] / m)
    # log p(z) and log q(z|x) sums of the share: oracle on the same 48 rows
    zc = z[rows].cpu().numpy()
    lp = onp.sample_log_likelihood(zc, reduce_batch_dim=False)
    lqx = onp.sample_log_likelihood(zc, mu[lo + rows].cpu().numpy(), lv[lo + rows].cpu().numpy(), reduce_batch_dim=False)
    from lvae import estimators as E
    glp = E.sample_log_likelihood(z[rows].contiguous(), reduce_batch_dim=False).double().cpu().numpy()
    glqx = E.sample_log_likelihood(z[rows].contiguous(), mu[lo + rows].contiguous(), lv[lo + rows].contiguous(),
                                   reduce_batch_dim=False).double().cpu().numpy()
    np.testing.assert_allclose(glp, lp, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(glqx, lqx, rtol=1e-5, atol=1e-3)


def _kernel_columns64(cols, pooled, alphas, eps, block=65536):
    """float64 K[r, c] = sum_alpha exp(-alpha (eps + |pooled_r - cols_c|^2)) for the given columns: [n, nc]."""
    c64 = cols.astype(np.float64)
    cn = (c64 ** 2).sum(1)
    out = np.empty((pooled.shape[0], cols.shape[0]))
    for lo in range(0, pooled.shape[0], block):
        p64 = pooled[lo:lo + block].astype(np.float64)
        d2 = eps + np.abs((p64 ** 2).sum(1)[:, None] + cn[None] - 2.0 * (p64 @ c64.T))
        k = np.zeros_like(d2)
        for a in alphas:
            k += np.exp(-float(np.float32(a)) * d2)        # the float32 values that cross the C ABI
        out[lo:lo + block] = k
    return out


def test_c4_kernel_columns_and_permutation_sums_vs_oracle():
    """configs[3]: n1 = n2 = 262,144, D = 32, the reference's 15 bandwidths (loss_and_optimisation.py:598), 100 + 1
    labelings.  The pooled 524,288^2 kernel matrix is never formed; 64 columns of it (straddling the two samples)
    are: lvae_kernel_sum_f32's per-column sums and lvae_mmd_perm_f32's q_p / lr_p / T shares for exactly those
    columns against float64 NumPy.  (MMDStatistic itself: parity unpinned, the oracle restates its algorithm.)"""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(13)
    n1 = n2 = 262144
    D, P, nc = 32, 100, 64
    alphas = [0.1 * i for i in range(15)]
    x = torch.randn(n1, D, device="cuda", generator=g)
    y = torch.randn(n2, D, device="cuda", generator=g) * 1.05
    pooled = torch.cat((x, y)).contiguous()
    n = n1 + n2
    lo = n1 - nc // 2
    cols = pooled[lo:lo + nc].contiguous()
    labels = E.permutation_labels(n1, n2, P, "cuda", g)                 # [P + 1, n], row 0 observed
    tot, cs = E.kernel_sum(cols, pooled, alphas, 1e-5, exclude_diag=True, diag_offset=lo, colsum=True)
    q, lr, T = E.mmd_permutation_sums(cols, lo, pooled, labels, alphas, 1e-5, symmetric=False)
    pooled_c, cols_c, lab = pooled.cpu().numpy(), cols.cpu().numpy(), labels.cpu().numpy().astype(np.float64)
    K = _kernel_columns64(cols_c, pooled_c, alphas, 1e-5)               # [n, nc], alpha = 0 adds 1 per pair
    K[lo + np.arange(nc), np.arange(nc)] = 0.0                          # excluded diagonal
    np.testing.assert_allclose(cs.cpu().numpy(), K.sum(0), rtol=5e-6)
    assert abs(float(tot) - K.sum()) <= 5e-6 * K.sum()
    Kp = K - 1.0                                                        # permutation sums leave alpha = 0 out (lvae.h)
    Kp[lo + np.arange(nc), np.arange(nc)] = 0.0
    KL = lab @ Kp                                                       # [P + 1, nc]: (K L_p)[c]
    Lc = lab[:, lo:lo + nc]
    r = Kp.sum(0)
    np.testing.assert_allclose(q.cpu().numpy(), (Lc * KL).sum(1), rtol=1e-4)
    np.testing.assert_allclose(lr.cpu().numpy(), (Lc * r[None]).sum(1), rtol=1e-5)
    assert abs(float(T) - r.sum()) <= 1e-5 * r.sum()


def test_c5_iw_sentences_vs_oracle():
    """configs[4]: B = 16,384 sentences x K = 1,000 samples, D = 32; 64 random sentences against
    oracle_c.reparam_iw (modules/encoder.py:148-157, loss_and_optimisation.py:93-101,:783-794)."""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(14)
    B, K, D = 16384, 1000, 32
    mu, lv = _posteriors(g, B, D)
    eps = torch.randn(B, K, D, device="cuda", generator=g)
    lpxz = 5.0 * torch.randn(B, K, device="cuda", generator=g) - 100.0
    out = E.reparam_iw(mu, lv, eps, lpxz, return_z=True)
    idx = torch.randperm(B, device="cuda", generator=g)[:64]
    z, lq, lp, iw = oracle_c.reparam_iw(mu[idx].cpu().numpy(), lv[idx].cpu().numpy(), eps[idx].cpu().numpy(),
                                        lpxz[idx].cpu().numpy())
    np.testing.assert_allclose(out["latent_z"][idx].cpu().numpy(), z, rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["log_q_z_x"][idx].cpu().numpy(), lq, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(out["log_p_z"][idx].cpu().numpy(), lp, rtol=1e-5, atol=1e-4)
    got = out["iw_ll"][idx].double().cpu().numpy()
    np.testing.assert_allclose(got, iw, rtol=1e-5, atol=1e-4)
    assert abs(got.mean() - iw.mean()) <= 1e-4 * abs(iw.mean())


def test_mi_bounds_k20000_d64_matrix_cores_vs_direct_form():
    """(regression: K = 20,000 at D = 64 used to be refused -- the workspace was sized for one of the two log q(z)
    plans of the call; the exact direct form is the reference here, the C oracle covers the same K at D = 32 below)"""
    from lvae import _native as N, mutual_information as MI
    g = torch.Generator(device="cuda").manual_seed(5)
    K, D = 20000, 64
    mu = torch.randn(K, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(K, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(K, D, device="cuda", generator=g) * torch.exp(0.25 * lv)
    auto = MI._bounds(z, (mu, torch.exp(0.5 * lv))).tolist()
    direct = MI._bounds(z, (mu, torch.exp(0.5 * lv)), path=N.PATH_DIRECT).tolist()
    for a, b in zip(auto, direct):
        assert abs(a - b) <= 1e-4 * abs(b), (auto, direct)


def test_mi_bounds_k20000_vs_oracle():
    """calc_upper_and_lower_bound_representational_mi (mutual_information.py:218-248) at K = 20,000 (the reference
    runs K = 128): both bounds against the C oracle over the full K x K matrix.  With quirk Q3 the critics'
    variance is std_j, i.e. "logvar" = 0.5 * logvar_j."""
    from lvae import mutual_information as MI
    rng = np.random.default_rng(15)
    K, D = 20000, 32
    mu = rng.standard_normal((K, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((K, D)) - 1).astype(np.float32)
    z = (mu + rng.standard_normal((K, D)).astype(np.float32) * np.exp(0.25 * lv)).astype(np.float32)
    mulv = np.concatenate([mu, lv], 1)
    glo, gup = MI.calc_upper_and_lower_bound_representational_mi(torch.from_numpy(z).cuda(), torch.from_numpy(mulv).cuda())
    lse_all, _ = oracle_c.log_qz(z, mu, 0.5 * lv, 0.0, False)            # LSE_j P[j, i], float64
    diag = onp.sample_log_likelihood(z, mu, 0.5 * lv, reduce_batch_dim=False)
    lower = float(np.mean(diag - (lse_all - math.log(K))))               # :165-171
    lse_off = oracle_c.log_qz_excl(z, mu, 0.5 * lv, 0.0, 0)              # the diagonal left out (:207)
    upper = float(np.mean(diag - (lse_off - math.log(K - 1))))           # :204-213
    assert abs(glo - lower) <= 1e-4 * abs(lower), (glo, lower)
    assert abs(gup - upper) <= 1e-4 * abs(upper), (gup, upper)
    assert glo <= math.log(K) + 1e-4 and gup >= glo - 1e-4
