"""GPU parity of the remaining hot-path rows against the reference's golden
vectors: analytic KL, MI bounds (Q3), q_z_estimate (Q2), _gaussian_log_likelihood, calc_mi_estimate,
GaussianKDE (Q6), the biased in-repo MMD, and -- parity unpinned -- the torch-two-sample MMD against the
NumPy restatement of its algorithm."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, rtol=1e-4, atol=1e-4):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", ["generic", "d64", "d20", "collapsed", "peaked"])
def test_kl_divergence(case):
    from lvae import estimators as E
    g = load_golden("logqz_" + case)
    mu, lv = dev(g["mu"]), dev(g["logvar"])
    for lam in (0.0, 0.5):
        kl, hk = E.kl_divergence(mu, lv, lam)
        close(kl, g[f"kl_lam{lam}"]); close(hk, g[f"hinge_kl_lam{lam}"])
        kl, hk = E.kl_divergence(mu, lv, lam, average_batch=False)
        close(kl, g[f"kl_rows_lam{lam}"], atol=1e-3); close(hk, g[f"hinge_kl_rows_lam{lam}"], atol=1e-3)
        kl, hk = E.kl_divergence(mu, lv, lam, average_batch=False, sum_latent=False)
        close(kl, g[f"kl_elem_lam{lam}"], atol=1e-4)
        amb = np.abs(g[f"kl_elem_lam{lam}"] - lam) < 1e-5     # float ties at the hinge threshold
        close(np.where(amb, 0, hk.cpu().numpy()), np.where(amb, 0, g[f"hinge_kl_elem_lam{lam}"]), atol=1e-4)
        kl, hk = E.kl_divergence(mu, lv, lam, average_batch=True, sum_latent=False)
        close(kl, g[f"kl_elem_lam{lam}"].mean(0), atol=1e-4)


@pytest.mark.parametrize("case", ["k128", "k16"])
def test_mi_bounds(case):
    from lvae import mutual_information as MI
    g = load_golden("mi_bounds_" + case)
    z, mulv = dev(g["z"]), dev(g["mu_logvars"])
    lower, upper = MI.calc_upper_and_lower_bound_representational_mi(z, mulv)
    assert abs(lower - float(g["lower"])) <= 1e-4 * abs(float(g["lower"]))
    assert abs(upper - float(g["upper"])) <= 1e-4 * abs(float(g["upper"]))
    D = g["mu_logvars"].shape[1] // 2
    mu, std = mulv[:, :D].contiguous(), torch.exp(0.5 * mulv[:, D:]).contiguous()
    P = MI.log_prob_pairs((mu, std), z)
    close(P, g["log_prob_pairs"], rtol=1e-4, atol=2e-3)
    assert abs(MI.mi_lower_bound(z, (mu, std)) - float(g["lower"])) <= 1e-4 * abs(float(g["lower"]))
    assert abs(MI.mi_upper_bound(z, (mu, std)) - float(g["upper"])) <= 1e-4 * abs(float(g["upper"]))
    # the reference's own calling convention: a list of torch distributions
    dists = [torch.distributions.MultivariateNormal(mu[i], torch.diag(std[i])) for i in range(min(8, len(mu)))]
    P8 = MI.log_prob_pairs(dists, z)
    close(P8, g["log_prob_pairs"][:len(dists)], rtol=1e-4, atol=2e-3)
    assert lower <= np.log(len(g["z"])) + 1e-5      # InfoNCE <= log K (float32 rounding slack)


@pytest.mark.parametrize("K,D", [(1500, 32), (700, 16), (900, 64), (600, 100), (500, 128), (333, 768), (290, 201)])
def test_mi_bounds_large_vs_oracle(K, D):
    """(every width of the pair kernel with the diagonal excluded: D = 16 / 32 / 64 / 128 variants, and the blocked
    direct kernel of wider latents -- 768 is the reference's module default, modules/encoder.py:9)"""
    from lvae import mutual_information as MI
    rng = np.random.default_rng(3)
    mu = rng.standard_normal((K, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((K, D)) - 1).astype(np.float32)
    z = (mu + rng.standard_normal((K, D)).astype(np.float32) * np.exp(0.25 * lv)).astype(np.float32)
    lo, up = onp.calc_upper_and_lower_bound_representational_mi(z, np.concatenate([mu, lv], 1))
    glo, gup = MI.calc_upper_and_lower_bound_representational_mi(dev(z), dev(np.concatenate([mu, lv], 1)))
    assert abs(glo - lo) <= 1e-4 * abs(lo) and abs(gup - up) <= 1e-4 * abs(up)


def test_q_z_estimate_gll_calc_mi():
    from lvae import mutual_information as MI
    g = load_golden("qz_estimate")
    z, mu, var = dev(g["z"]), dev(g["mu"]), dev(g["var"])
    q = MI.q_z_estimate(z, mu, var)
    assert tuple(q.shape) == g["q_z_estimate"].shape == (1, g["mu"].shape[0])     # Q2
    close(q, g["q_z_estimate"], rtol=1e-4, atol=2e-3)
    one = torch.tensor([[1.]]).cuda()
    close(MI._gaussian_log_likelihood(z.unsqueeze(1), one, 2, mu=mu, var=var), g["gll_full"], atol=2e-3)
    close(MI._gaussian_log_likelihood(z.unsqueeze(1), one, 2, mu=None, var=var), g["gll_nomu"], atol=2e-3)
    close(MI._gaussian_log_likelihood(z.unsqueeze(1), one, 2, mu=mu, var=None), g["gll_novar"], atol=2e-3)
    close(MI._gaussian_log_likelihood(z.unsqueeze(1), one, 2), g["gll_std"], atol=2e-3)
    for m in ("zhao", "hoffman"):
        mi, mkl = MI.calc_mi_estimate(dev(g["cme_log_q_z_x"]), None, None, None, dev(g["cme_log_p_z"]),
                                      float(g["cme_avg_kl"]), mi_method=m)
        assert abs(mi - float(g[f"cme_mi_{m}"])) <= 1e-4 * abs(float(g[f"cme_mi_{m}"]))
        assert abs(mkl - float(g[f"cme_marg_kl_{m}"])) <= 1e-4 * abs(float(g[f"cme_marg_kl_{m}"]))


def test_q_z_estimate_wide_latents():
    """D > 128: explicit matrix + row logsumexp (quirk Q2 kept: one value per POSTERIOR)."""
    from lvae import mutual_information as MI
    rng = np.random.default_rng(8)
    S, N_, D = 90, 130, 200
    mu = rng.standard_normal((N_, D)).astype(np.float32)
    var = np.exp(0.4 * rng.standard_normal((N_, D)) - 1).astype(np.float32)
    z = (mu[rng.integers(0, N_, S)] + rng.standard_normal((S, D)) * 0.6).astype(np.float32)
    q = MI.q_z_estimate(dev(z), dev(mu), dev(var))
    want = onp.q_z_estimate(z, mu, var)
    assert tuple(q.shape) == (1, N_)
    close(q, np.asarray(want).reshape(1, N_), rtol=1e-4, atol=2e-3)


def _kde_stable64(X, Y, bw):
    """float64, max-stabilised evaluation of the KDE score (finite for any query)."""
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    n, d = X.shape
    out = np.empty(Y.shape[0])
    for lo in range(0, Y.shape[0], 512):
        u = (X[:, None, :] - Y[None, lo:lo + 512, :]) / bw
        logk = -0.5 * (u ** 2).sum(axis=2) - 0.5 * d * np.log(2 * np.pi)
        out[lo:lo + 512] = onp._logsumexp(logk, 0) - np.log(n) - d * np.log(bw)
    return out


def test_kde():
    from lvae import estimators as E
    from lvae.kde import GaussianKDE
    g = load_golden("kde")
    lat = dev(g["latents"])
    flat = lat.reshape(-1, 1)
    k = GaussianKDE(X=flat, bw="scott")
    assert abs(k.bw - float(g["bw_scott_1d"])) < 1e-12
    close(k.score_samples(flat), g["score_1d_scott"], atol=1e-4)
    k2 = GaussianKDE(X=flat, bw=0.25)
    s = k2.score_samples(dev(g["Y_1d"])).cpu().numpy()
    assert np.isneginf(s[-3:]).all()                                   # Q6
    close(s[:-3], g["score_1d_bw025"][:-3], atol=1e-4)
    k3 = GaussianKDE(X=dev(g["X3"]), bw="scott")
    close(k3.score_samples(dev(g["Y3"])), g["score_3d_scott"], atol=1e-4)
    got = E.get_1d_kde_dim_marg_kl_estimate(lat, dev(g["log_p_z_mean"]))
    assert abs(float(got) - float(g["kde1d_dim_marg_kl"])) <= 1e-4 * abs(float(g["kde1d_dim_marg_kl"]))
    # stabilised variant (no counterpart in the reference): same values where the reference is finite, and
    # finite values -- checked against the float64 oracle -- where the reference underflows to -inf
    s2 = k2.score_samples(dev(g["Y_1d"]), stabilise=True).cpu().numpy()
    close(s2[:-3], g["score_1d_bw025"][:-3], atol=1e-4)
    ref64 = _kde_stable64(g["latents"].reshape(-1, 1), g["Y_1d"], 0.25)
    assert np.isfinite(s2).all()
    np.testing.assert_allclose(s2, ref64, rtol=1e-4)
    close(k3.score_samples(dev(g["Y3"]), stabilise=True), g["score_3d_scott"], atol=1e-4)


def test_kde_dataset_scale_vs_oracle():
    """Ragged sizes that exercise the split of the fit points and the packed 1-D tail."""
    from lvae.kde import GaussianKDE
    rng = np.random.default_rng(4)
    X = rng.standard_normal((20011, 1)).astype(np.float32)
    Y = (1.5 * rng.standard_normal((3001, 1))).astype(np.float32)
    for stab in (False, True):
        got = GaussianKDE(X=dev(X), bw="scott").score_samples(dev(Y), stabilise=stab).cpu().numpy()
        bw = X.shape[0] ** (-1.0 / 5.0)
        np.testing.assert_allclose(got, _kde_stable64(X, Y, bw), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", ["n256", "n100_d64"])
def test_biased_mmd(case):
    from lvae import estimators as E
    g = load_golden("mmd_" + case)
    x, y = dev(g["x"]), dev(g["y"])
    m = g["kxy_block"].shape[0]
    close(E.gaussian_kernel(x[:m], y[:m]), g["kxy_block"], atol=1e-6)
    n = x.shape[0]
    out = E.mmd_rbf(x, y, [1.0 / x.shape[1] ** 2], 0.0, unbiased=False).cpu().numpy()
    # component means at 1e-6; the statistic is a difference of O(1) means: absolute tolerance
    # against the component magnitude
    close(out[1] / n ** 2, g["kxx_mean"], atol=2e-6)
    close(out[2] / n ** 2, g["kyy_mean"], atol=2e-6)
    close(out[3] / n ** 2, g["kxy_mean"], atol=2e-6)
    close(E.maximum_mean_discrepancy(y, x), g["mmd"], rtol=0, atol=2e-6)
    ref64 = onp.maximum_mean_discrepancy(g["y"], g["x"])
    assert abs(out[0] - ref64) <= 1e-4 * abs(ref64) + 1e-7


@pytest.mark.parametrize("n1,n2,D", [(300, 300, 32), (257, 190, 64), (1000, 1200, 32)])
def test_tts_mmd_unpinned(n1, n2, D):
    """torch-two-sample is absent from the reference tree: PARITY UNPINNED.  Checked against the float64
    NumPy restatement of its published algorithm (oracle_np.mmd_statistic_tts)."""
    from lvae.two_sample import MMDStatistic
    rng = np.random.default_rng(n1 + D)
    a = rng.standard_normal((n1, D)).astype(np.float32)
    b = (rng.standard_normal((n2, D)) * 0.8 + 0.3).astype(np.float32)
    alphas = [0.1 * i for i in range(15)]
    ref, K = onp.mmd_statistic_tts(a, b, alphas, ret_matrix=True)
    got = float(MMDStatistic(n1, n2)(dev(a), dev(b), alphas))
    assert abs(got - ref) <= 1e-4 * abs(ref)
    if n1 <= 300:
        got2, Kg = MMDStatistic(n1, n2)(dev(a), dev(b), alphas, ret_matrix=True)
        close(Kg, K, rtol=1e-4, atol=1e-5)
    # multiples without the pairing take the general polynomial
    al4 = [0.0, 0.1, 0.2, 0.3, 0.7, 1.1]
    ref4 = onp.mmd_statistic_tts(a, b, al4)
    got4 = float(MMDStatistic(n1, n2)(dev(a), dev(b), al4))
    assert abs(got4 - ref4) <= 1e-4 * abs(ref4)
    # non-multiple bandwidths take the general (one exp per bandwidth) epilogue
    al2 = [0.05, 0.13, 0.31]
    ref2 = onp.mmd_statistic_tts(a, b, al2)
    got2 = float(MMDStatistic(n1, n2)(dev(a), dev(b), al2))
    assert abs(got2 - ref2) <= 1e-4 * abs(ref2)
    # same distribution: statistic ~ 0 (may be slightly negative), error judged against the kernel scale
    c = rng.standard_normal((n2, D)).astype(np.float32)
    ref3 = onp.mmd_statistic_tts(a, c, alphas)
    got3 = float(MMDStatistic(n1, n2)(dev(a), dev(c), alphas))
    assert abs(got3 - ref3) <= 2e-6 * 15


@pytest.mark.parametrize("na,nb,D,off", [(300, 500, 32, 120), (257, 257, 64, 0), (90, 700, 16, -10),
                                         (130, 64, 128, 40)])
def test_kernel_sum_blocks(na, nb, D, off):
    """Row-block kernel sums (the multi-GPU MMD building block) against float64 NumPy, with the
    excluded diagonal at an offset, alpha = 0 bandwidths, and per-row sums."""
    from lvae import estimators as E
    rng = np.random.default_rng(na + nb)
    a = rng.standard_normal((na, D)).astype(np.float32)
    b = (0.9 * rng.standard_normal((nb, D)) + 0.2).astype(np.float32)
    if off >= 0 and na + off <= nb:
        b[off:off + na] = a                 # the excluded pairs are genuine self-pairs
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    # (paired-coefficient polynomial, one bandwidth, arbitrary bandwidths, zeros only, general polynomial,
    #  paired polynomial with unequal pairs)
    for alphas, eps in (([0.1 * i for i in range(15)], 1e-5), ([1.0 / D ** 2], 0.0), ([0.0, 0.07, 0.19], 1e-5),
                        ([0.0, 0.0], 0.0), ([0.05, 0.1, 0.15, 0.25, 0.25, 0.8], 1e-5),
                        ([0.03, 0.06, 0.09, 0.09, 0.12, 0.12, 0.45, 0.48], 1e-5),        # (degree 16: general form)
                        ([0.01, 0.02, 0.03, 0.03, 0.04, 0.04, 0.13, 0.14], 1e-5)):       # (degree 14, pairs 1, 2, 0.., 1)
        d2 = eps + np.abs((a64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None] - 2.0 * a64 @ b64.T)
        K = sum(np.exp(-al * d2) for al in alphas)
        for excl in (False, True):
            Kx = K.copy()
            if excl:
                i = np.arange(na)
                j = i + off
                ok = (j >= 0) & (j < nb)
                Kx[i[ok], j[ok]] = 0.0
            tot, cs = E.kernel_sum(dev(a), dev(b), alphas, eps, excl, off, colsum=True)
            assert abs(float(tot) - Kx.sum()) <= 2e-6 * Kx.sum(), (alphas, excl)
            np.testing.assert_allclose(cs.cpu().numpy(), Kx.sum(1), rtol=1e-5)
            tot2 = E.kernel_sum(dev(a), dev(b), alphas, eps, excl, off)
            assert float(tot2) == float(tot)
    # symmetric evaluation: mirrored 32-aligned chunks of one sample against itself add up to the full sum
    from lvae.distributed import mirrored_chunks
    alphas = [0.1 * i for i in range(15)]
    b64 = b.astype(np.float64)
    d2 = 1e-5 + np.abs((b64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None] - 2.0 * b64 @ b64.T)
    K = sum(np.exp(-al * d2) for al in alphas)
    for excl in (False, True):
        want = K.sum() - (np.trace(K) if excl else 0.0)
        acc = 0.0
        for rank in range(3):
            for lo, hi in mirrored_chunks(nb, 3, rank):
                acc += float(E.kernel_sum(dev(b[lo:hi]), dev(b), alphas, 1e-5, excl, lo, symmetric=True))
        # (with the diagonal included, the self-pairs' d2 = |x|^2 - 2 x.x + |x|^2 is fp32 rounding noise, not 0)
        assert abs(acc - want) <= (2e-6 if excl else 2e-5) * want, excl


@pytest.mark.parametrize("nsplit", [1, 3, 5, 10, 20])
def test_permutation_sums_symmetric_walk_for_every_split_length(nsplit, lvae_env):
    """The upper-triangle walk (weights 0 / 1 / 2 per tile, different for the four waves of a workgroup) against
    the full walk, with the posterior axis cut into ranges of 20, 7, 4, 2 and 1 blocks (LVAE_NSPLIT overrides the
    plan): label sums of tiles below the diagonal must not leak in, whatever the range length."""
    import os
    from lvae import estimators as E
    lvae_env.set("LVAE_NSPLIT", str(nsplit))
    rng = np.random.default_rng(nsplit)
    n, D, P = 640, 32, 37
    x = dev(rng.standard_normal((n, D)).astype(np.float32))
    labels = E.permutation_labels(n // 2, n - n // 2, P, "cuda", torch.Generator(device="cuda").manual_seed(1))
    alphas = [0.1 * i for i in range(15)]
    q, lr, tot = E.mmd_permutation_sums(x, 0, x, labels, alphas, 1e-5, True)
    qf, lrf, totf = E.mmd_permutation_sums(x, 0, x, labels, alphas, 1e-5, False)
    np.testing.assert_allclose(q.cpu().numpy(), qf.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(lr.cpu().numpy(), lrf.cpu().numpy(), rtol=1e-6)
    assert abs(float(tot) - float(totf)) <= 1e-6 * float(totf)
    assert os.environ["LVAE_NSPLIT"] == str(nsplit)


@pytest.mark.parametrize("n1,n2,D,P", [(300, 340, 32, 20), (257, 190, 64, 130), (1000, 1500, 16, 9),
                                         (200, 150, 128, 5), (2048, 2048, 32, 33), (130, 110, 250, 7)])
def test_mmd_permutation_test_unpinned(n1, n2, D, P):
    """Permutation statistics (PARITY UNPINNED, like the statistic itself) against the float64 NumPy
    restatement evaluated on the dense kernel matrix.  The statistic is a difference of O(1) means, so the
    error is judged against the kernel scale, and the observed statistic against 1e-4."""
    from lvae import estimators as E
    rng = np.random.default_rng(n1 + D + P)
    scale = 1.0 if D <= 128 else 4.0 / np.sqrt(D)        # keep the kernel values away from underflow at wide D
    a = (rng.standard_normal((n1, D)) * scale).astype(np.float32)
    b = ((rng.standard_normal((n2, D)) * 0.9 + 0.15) * scale).astype(np.float32)
    # (fifteen multiples: the paired-coefficient polynomial; D = 16: a general polynomial; D = 64: one bandwidth)
    alphas = [0.1 * i for i in range(15)] if D not in (16, 64) else ([0.02] if D == 64 else [0.1, 0.2, 0.3, 0.5, 0.5, 0.9])
    ref_obs, K = onp.mmd_statistic_tts(a, b, alphas, ret_matrix=True)
    labels = E.permutation_labels(n1, n2, P, "cuda", torch.Generator(device="cuda").manual_seed(5))
    assert labels.shape == (P + 1, n1 + n2) and bool((labels.sum(1) == n1).all())
    lab_np = labels.cpu().numpy()
    ref = onp.mmd_permutation_stats(K, n1, n2, 1 - lab_np)      # oracle convention: 0 -> sample 1
    out = E.mmd_permutation_test(dev(a), dev(b), alphas, labels=labels)
    got = np.concatenate([[float(out["statistic"])], out["statistics"].cpu().numpy()])
    assert abs(ref[0] - ref_obs) <= 1e-12 * max(1.0, abs(ref_obs))
    assert abs(got[0] - ref_obs) <= 1e-4 * abs(ref_obs)
    kscale = len([x for x in alphas if x > 0])
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6 * kscale)
    want_p = float((ref[1:] >= ref[0]).mean())
    assert abs(float(out["pval"]) - want_p) <= 1.0 / P + 1e-12
    # the same through the torch-two-sample style object
    from lvae.two_sample import MMDStatistic
    st = MMDStatistic(n1, n2)
    st(dev(a), dev(b), alphas)
    pv = st.pval(None, n_permutations=16)
    assert 0.0 <= pv <= 1.0
    if n1 + n2 <= 700:       # upstream's calling convention: the matrix of THAT call is accepted, any other one refused
        with pytest.raises(ValueError, match="ret_matrix"):
            st.pval(torch.zeros(n1 + n2, n1 + n2, device="cuda"), n_permutations=4)
        _, mat = st(dev(a), dev(b), alphas, ret_matrix=True)
        assert 0.0 <= st.pval(mat, n_permutations=8) <= 1.0
        with pytest.raises(ValueError, match="ret_matrix"):
            st.pval(mat.clone(), n_permutations=4)
    # column-sharded evaluation adds up to the same sums, with the full rows ...
    pooled = torch.cat((dev(a), dev(b)), 0)
    q, lr, tot = E.mmd_permutation_sums(pooled, 0, pooled, labels, alphas, 1e-5)         # symmetric
    qf, lrf, totf = E.mmd_permutation_sums(pooled, 0, pooled, labels, alphas, 1e-5, symmetric=False)
    np.testing.assert_allclose(qf.cpu().numpy(), q.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(lrf.cpu().numpy(), lr.cpu().numpy(), rtol=1e-6)
    assert abs(float(totf) - float(tot)) <= 1e-6 * float(tot)      # K[r][c] and K[c][r] round differently
    cut = (n1 + n2) // 3
    q1, lr1, t1 = E.mmd_permutation_sums(pooled[:cut].contiguous(), 0, pooled, labels, alphas, 1e-5)
    q2, lr2, t2 = E.mmd_permutation_sums(pooled[cut:].contiguous(), cut, pooled, labels, alphas, 1e-5)
    np.testing.assert_allclose((q1 + q2).cpu().numpy(), q.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose((lr1 + lr2).cpu().numpy(), lr.cpu().numpy(), rtol=1e-6)
    assert abs(float(t1 + t2) - float(tot)) <= 1e-6 * float(tot)
    # ... and with the upper triangle only (32-aligned mirrored chunks, as the sharded test uses them)
    from lvae.distributed import mirrored_chunks
    acc = None
    for rank in range(3):
        for lo, hi in mirrored_chunks(n1 + n2, 3, rank):
            part = E.mmd_permutation_sums(pooled[lo:hi].contiguous(), lo, pooled, labels, alphas, 1e-5, True)
            vec = torch.cat((part[0], part[1], part[2].reshape(1)))
            acc = vec if acc is None else acc + vec
    want = torch.cat((q, lr, tot.reshape(1)))
    np.testing.assert_allclose(acc.cpu().numpy(), want.cpu().numpy(), rtol=1e-6)


def test_get_tts_mmmd_and_encode_tail():
    from lvae import encoder_ops as EO
    from lvae import estimators as E
    g = load_golden("iw_k37_d64")
    mu, lv, eps = dev(g["mu"]), dev(g["logvar"]), dev(g["eps"])
    z = EO.reparameterize(mu, lv, n_samples=eps.shape[1], eps=eps)
    close(z, g["z"], rtol=1e-6, atol=1e-6)
    assert tuple(EO.reparameterize(mu, lv, n_samples=3).shape) == (mu.shape[0], 3, mu.shape[1])
    assert tuple(EO.sample_from_prior(32, 5, "cuda:0").shape) == (5, 32)
    out = EO.encode_tail(mu, lv, n_samples=eps.shape[1], eps=eps)
    close(out["log_q_z_x"], g["log_q_z_x"], atol=1e-3)
    close(out["log_p_z"], g["log_p_z"], atol=1e-3)
    ref_lq, ref_pm = onp.approximate_log_q_z(g["mu"], g["logvar"], g["z"][:, 0], "chen", 10000, True, True)
    assert abs(float(out["log_q_z"]) - ref_lq) <= 1e-4 * abs(ref_lq)
    assert abs(float(out["log_q_z_prod_marg"]) - ref_pm) <= 1e-4 * abs(ref_pm)
    prior = torch.randn(mu.shape[0], mu.shape[1], device="cuda")
    t = E.get_tts_mmmd(z[:, :1, :], prior)
    ref = onp.mmd_statistic_tts(g["z"][:, 0], prior.cpu().numpy(), [0.1 * i for i in range(15)])
    assert abs(float(t) - ref) <= 1e-4 * abs(ref) + 1e-5
    # the intended marginal KL and the reference's raising behaviour (Q1)
    kl = E.approximate_marginal_KL(mu, lv, z[:, 0].contiguous(), "chen", 10000)
    ref_kl = onp.approximate_marginal_KL_intended(g["mu"], g["logvar"], g["z"][:, 0], "chen", 10000)
    assert abs(float(kl) - ref_kl) <= 1e-4 * abs(ref_kl)
    with pytest.raises(AttributeError):
        E.approximate_marginal_KL(mu, lv, z[:, 0].contiguous(), "chen", 10000, strict=True)
    tc = E.approximate_total_correlation(mu, lv, z[:, 0].contiguous(), "chen", 10000)
    assert abs(float(tc) - (ref_lq - ref_pm)) <= 1e-4 * abs(ref_lq - ref_pm) + 1e-3


def test_mmd_full_size_properties():
    """BASELINE configs[3] at full size (N = M = 262,144, D = 32, 100 permutations), through properties:
    the statistic is symmetric in its two samples, the permutation machinery reproduces it for the observed
    labeling, relabelings of two samples from ONE distribution scatter around it (p-value not extreme), and a
    shifted second sample is detected (p-value 0)."""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(9)
    n, D, P = 262144, 32, 100
    alphas = [0.1 * i for i in range(15)]
    x = torch.randn(n, D, device="cuda", generator=g)
    y = torch.randn(n, D, device="cuda", generator=g)
    s_xy = float(E.mmd_rbf(x, y, alphas, 1e-5, unbiased=True)[0])
    s_yx = float(E.mmd_rbf(y, x, alphas, 1e-5, unbiased=True)[0])
    assert abs(s_xy - s_yx) <= 1e-7
    labels = E.permutation_labels(n, n, P, "cuda", g)
    out = E.mmd_permutation_test(x, y, alphas, labels=labels)
    assert abs(float(out["statistic"]) - s_xy) <= 1e-7
    st = out["statistics"].cpu().numpy()
    assert 0.02 <= float(out["pval"]) <= 0.98, float(out["pval"])
    assert abs(s_xy) <= 6.0 * st.std() + abs(st.mean())
    out2 = E.mmd_permutation_test(x, y + 0.02, alphas, labels=labels)
    assert float(out2["pval"]) == 0.0 and float(out2["statistic"]) > st.max()


@pytest.mark.parametrize("n1,n2,D", [(64, 64, 768), (150, 97, 200), (700, 333, 129)])
def test_mmd_wide_latents(n1, n2, D):
    """Wide latents: up to D = 256 the kernel sums stay on the matrix cores (two dimensions per MFMA step), beyond
    (e.g. a 768-dimensional latent) they take the direct-difference kernel; statistic, biased MMD, row-block
    sums with an excluded diagonal and column sums against float64 NumPy."""
    from lvae import estimators as E
    from lvae.two_sample import MMDStatistic
    rng = np.random.default_rng(n1 + D)
    a = (rng.standard_normal((n1, D)) / np.sqrt(D) * 4).astype(np.float32)
    b = (rng.standard_normal((n2, D)) / np.sqrt(D) * 4 + 0.02).astype(np.float32)
    alphas = [0.1 * i for i in range(15)]
    ref = onp.mmd_statistic_tts(a, b, alphas)
    got = float(MMDStatistic(n1, n2)(dev(a), dev(b), alphas))
    assert abs(got - ref) <= 2e-6 * 15
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    for al, eps in (([1.0 / D], 0.0), ([0.0, 0.07, 0.19], 1e-5)):
        d2 = eps + ((a64[:, None, :] - b64[None]) ** 2).sum(2)
        K = sum(np.exp(-x * d2) for x in al)
        tot, cs = E.kernel_sum(dev(a), dev(b), al, eps, False, 0, colsum=True)
        assert abs(float(tot) - K.sum()) <= 2e-6 * K.sum()
        np.testing.assert_allclose(cs.cpu().numpy(), K.sum(1), rtol=1e-5)
    # excluded diagonal at an offset; a symmetric request is served as a plain one by the direct-difference
    # kernel (D > 256) -- on the matrix cores it returns the owner's share of the upper-triangle walk instead
    d2 = 1e-5 + ((b64[:, None, :] - b64[None]) ** 2).sum(2)
    K = sum(np.exp(-x * d2) for x in alphas)
    lo, hi = 32, min(n2, 96)
    want = K[lo:hi].sum() - np.trace(K[lo:hi, lo:hi])
    for sym in ((False, True) if D > 256 else (False,)):
        got = float(E.kernel_sum(dev(b[lo:hi]), dev(b), alphas, 1e-5, True, lo, symmetric=sym))
        assert abs(got - want) <= 2e-6 * want, sym
    # the biased MMD of the encode tail (maximum_mean_discrepancy, loss_and_optimisation.py:46-60)
    z = dev(a)
    x = torch.randn(n1, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    got = float(E.maximum_mean_discrepancy(z, prior_sample=x))
    want = onp.maximum_mean_discrepancy(a, x.cpu().numpy())
    assert abs(got - want) <= 1e-4 * abs(want) + 1e-7


def test_kde_batch_is_normalised_by_the_fit_size():
    """score_samples(Y, X=batch) divides by the size of the FIT set (utils_latent_analysis_optimisation.py:58)."""
    from lvae.kde import GaussianKDE
    g = load_golden("kde_batch")
    X, Y = dev(g["X"]), dev(g["Y"])
    k = GaussianKDE(X=X, bw="scott")
    assert abs(k.bw - float(g["bw"])) < 1e-12
    close(k.score_samples(Y), g["score_full"], atol=1e-4)
    close(k.score_samples(Y, X=X[:128]), g["score_batch_128"], atol=1e-4)
    close(k.score_samples(Y, X=X[300:]), g["score_batch_tail"], atol=1e-4)
    close(k.score_samples(Y, X=X[300:], stabilise=True), g["score_batch_tail"], atol=1e-4)
    k2 = GaussianKDE(X=dev(g["X2"]), bw=0.4)
    close(k2.score_samples(dev(g["Y2"]), X=dev(g["X2"])[:77]), g["score2_batch_77"], atol=1e-4)


def test_gaussian_log_likelihood_generic_masks_and_axes():
    """_gaussian_log_likelihood away from q_z_estimate's layout: real masks, other axes, 2-D samples
    (mutual_information.py:26-41), against values the reference produced."""
    from lvae import mutual_information as MI
    g = load_golden("gll_generic")
    f = MI._gaussian_log_likelihood
    z3, mu, var = dev(g["z"]).unsqueeze(1), dev(g["mu"]), dev(g["var"])
    msn, msd, mnd = dev(g["mask_sn"]), dev(g["mask_sd"]), dev(g["mask_nd"])
    for got, key in ((f(z3, msn, 2, mu=mu, var=var), "a_full_dim2"), (f(z3, msn, 2, mu=None, var=var), "a_nomu_dim2"),
                     (f(z3, msn, 2, mu=mu, var=None), "a_novar_dim2"), (f(z3, msn[:, :1], 2), "a_std_dim2"),
                     (f(z3, msd, 1, mu=mu, var=var), "b_full_dim1"), (f(z3, mnd, 0, mu=mu, var=var), "c_full_dim0")):
        assert tuple(got.shape) == g[key].shape, key
        close(got, g[key], rtol=1e-4, atol=2e-3)
    s2, mu2, var2, mb = dev(g["s2"]), dev(g["mu2"]), dev(g["var2"]), dev(g["mask_b"])
    for got, key in ((f(s2, mb, 1, mu=mu2, var=var2), "d_full_dim1"), (f(s2, mb, 1), "d_std_dim1"),
                     (f(s2, mb, -1, mu=mu2, var=None), "d_novar_dimm1")):
        assert tuple(got.shape) == g[key].shape, key
        close(got, g[key], rtol=1e-4, atol=1e-3)
