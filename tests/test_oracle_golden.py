"""Pins the CPU oracle (oracle/oracle_np.py) to the reference: every function is compared with the
golden vectors the reference's own code produced (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle_np as onp

LOGQZ_CASES = ["generic", "ragged", "d64", "collapsed", "near_collapsed", "peaked", "far", "d20",
               "d16", "n1"]
METHODS = [("chen10000", "chen", 10000), ("chen42068", "chen", 42068), ("all", "all", None)]


def close(a, b, rtol=2e-5, atol=2e-5):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                               rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", LOGQZ_CASES)
def test_approximate_log_q_z(case):
    g = load_golden("logqz_" + case)
    # the reference computes in float32; on the peaked case its own rounding noise is ~1e-3 abs
    tol = dict(rtol=1e-4, atol=2e-3) if case == "peaked" else dict(rtol=2e-5, atol=1e-4)
    for tag, method, ds in METHODS:
        lq, lqpm = onp.approximate_log_q_z(g["mu"], g["logvar"], g["z"], method=method, dataset_size=ds,
                                           prod_marginals=True, reduce_mean=False)
        close(lq, g[f"log_q_z_{tag}"], **tol)
        close(lqpm, g[f"log_q_z_prod_marg_{tag}"], **tol)
        close(lq, g[f"log_q_z_noprod_{tag}"], **tol)
        a, b = onp.approximate_log_q_z(g["mu"], g["logvar"], g["z"], method=method, dataset_size=ds,
                                       prod_marginals=True, reduce_mean=True)
        close(a, g[f"log_q_z_mean_{tag}"], **tol)
        close(b, g[f"log_q_z_prod_marg_mean_{tag}"], **tol)
    close(onp.approximate_total_correlation(g["mu"], g["logvar"], g["z"], "chen", 42068),
          g["total_correlation_chen42068"], rtol=1e-4, atol=2e-3)


def test_q1_reduce_mean_without_prod_marginals_raises():
    g = load_golden("logqz_generic")
    assert int(g["marginal_kl_raises"]) == 1
    with pytest.raises(AttributeError):
        onp.approximate_log_q_z(g["mu"], g["logvar"], g["z"], prod_marginals=False, reduce_mean=True)


@pytest.mark.parametrize("case", LOGQZ_CASES)
def test_sample_log_likelihood_and_kl(case):
    g = load_golden("logqz_" + case)
    tol = dict(rtol=1e-4, atol=2e-3) if case in ("peaked", "far") else dict(rtol=2e-5, atol=1e-4)
    close(onp.sample_log_likelihood(g["z"], reduce_batch_dim=False), g["log_p_z"], **tol)
    close(onp.sample_log_likelihood(g["z"], reduce_latent_dim=False, reduce_batch_dim=False),
          g["log_p_z_nored"], **tol)
    close(onp.sample_log_likelihood(g["z"]), g["log_p_z_batchmean"], **tol)
    if "log_q_z_x" in g:
        close(onp.sample_log_likelihood(g["z"], g["mu"], g["logvar"], reduce_batch_dim=False),
              g["log_q_z_x"], **tol)
        close(onp.sample_log_likelihood(g["z"], g["mu"], g["logvar"], False, False),
              g["log_q_z_x_nored"], **tol)
    for lam in (0.0, 0.5):
        kl, hk = onp.kl_divergence(g["mu"], g["logvar"], lam)
        close(kl, g[f"kl_lam{lam}"]); close(hk, g[f"hinge_kl_lam{lam}"])
        kl, hk = onp.kl_divergence(g["mu"], g["logvar"], lam, average_batch=False)
        close(kl, g[f"kl_rows_lam{lam}"]); close(hk, g[f"hinge_kl_rows_lam{lam}"])
        kl, hk = onp.kl_divergence(g["mu"], g["logvar"], lam, average_batch=False, sum_latent=False)
        close(kl, g[f"kl_elem_lam{lam}"], atol=1e-5)
        # the hinge mask flips on float32-vs-float64 ties at the threshold; compare where unambiguous
        amb = np.abs(g[f"kl_elem_lam{lam}"] - lam) < 1e-5
        close(np.where(amb, 0, hk), np.where(amb, 0, g[f"hinge_kl_elem_lam{lam}"]), atol=1e-5)


@pytest.mark.parametrize("case", ["k128", "k16"])
def test_mi_bounds(case):
    g = load_golden("mi_bounds_" + case)
    D = g["mu_logvars"].shape[1] // 2
    mu, logvar = g["mu_logvars"][:, :D], g["mu_logvars"][:, D:]
    P = onp.log_prob_pairs(g["z"], mu, np.exp(0.5 * logvar.astype(np.float64)))   # Q3
    close(P, g["log_prob_pairs"], rtol=2e-5, atol=2e-3)
    lo, up = onp.calc_upper_and_lower_bound_representational_mi(g["z"], g["mu_logvars"])
    close(lo, g["lower"], rtol=1e-5, atol=1e-5)
    close(up, g["upper"], rtol=1e-5, atol=1e-4)
    assert lo <= np.log(P.shape[0]) + 1e-9


def test_q_z_estimate_and_gll_and_calc_mi():
    g = load_golden("qz_estimate")
    close(onp.gaussian_log_likelihood(g["z"], g["mu"], g["var"]), g["gll_full"], atol=2e-4)
    close(onp.gaussian_log_likelihood(g["z"], None, g["var"]), g["gll_nomu"], atol=2e-4)
    close(onp.gaussian_log_likelihood(g["z"], g["mu"], None), g["gll_novar"], atol=2e-4)
    close(onp.gaussian_log_likelihood(g["z"]), g["gll_std"], atol=2e-4)
    q = onp.q_z_estimate(g["z"], g["mu"], g["var"])
    assert q.shape == g["q_z_estimate"].shape == (1, g["mu"].shape[0])           # Q2
    close(q, g["q_z_estimate"], atol=2e-4)
    for m in ("zhao", "hoffman"):
        mi, mkl = onp.calc_mi_estimate(g["cme_log_q_z_x"], g["cme_log_p_z"], g["cme_avg_kl"], m)
        close(mi, g[f"cme_mi_{m}"], atol=1e-4)
        close(mkl, g[f"cme_marg_kl_{m}"], atol=1e-4)


@pytest.mark.parametrize("case", ["k1", "k600", "k37_d64"])
def test_iw(case):
    g = load_golden("iw_" + case)
    z = onp.reparameterize(g["mu"], g["logvar"], g["eps"])
    close(z, g["z"], rtol=1e-6, atol=1e-6)
    lq = onp.sample_log_likelihood(g["z"], g["mu"], g["logvar"], reduce_batch_dim=False)
    lp = onp.sample_log_likelihood(g["z"], reduce_batch_dim=False)
    close(lq, g["log_q_z_x"], atol=1e-4)
    close(lp, g["log_p_z"], atol=1e-4)
    close(onp.iw_log_p_x(g["log_p_x_z"], g["log_p_z"], g["log_q_z_x"]), g["iw_ll"], atol=1e-4)
    if case == "k1":   # K = 1: the IW estimate is the single log-weight
        close(g["iw_ll"], g["log_p_x_z"][:, 0] + g["log_p_z"][:, 0] - g["log_q_z_x"][:, 0], atol=1e-4)


def test_kde():
    g = load_golden("kde")
    lat = g["latents"]
    flat = lat.reshape(-1, 1)
    bw = onp.scott_bandwidth(flat.shape[0], 1)
    close(bw, g["bw_scott_1d"], rtol=1e-12, atol=0)
    close(onp.kde_score_samples(flat, flat, bw), g["score_1d_scott"], atol=1e-5)
    s = onp.kde_score_samples(flat, g["Y_1d"], 0.25)
    assert np.isneginf(g["score_1d_bw025"][-3:]).all() and np.isneginf(s[-3:]).all()   # Q6
    close(s[:-3], g["score_1d_bw025"][:-3], atol=1e-5)
    bw3 = onp.scott_bandwidth(*g["X3"].shape)
    close(bw3, g["bw_scott_3d"], rtol=1e-12, atol=0)
    close(onp.kde_score_samples(g["X3"], g["Y3"], bw3), g["score_3d_scott"], atol=1e-5)
    close(onp.get_1d_kde_dim_marg_kl_estimate(lat, g["log_p_z_mean"]), g["kde1d_dim_marg_kl"],
          atol=1e-4)


@pytest.mark.parametrize("case", ["n256", "n100_d64"])
def test_biased_mmd(case):
    g = load_golden("mmd_" + case)
    m = g["kxy_block"].shape[0]
    close(onp.gaussian_kernel(g["x"][:m], g["y"][:m]), g["kxy_block"], atol=1e-6)
    close(onp.gaussian_kernel(g["x"], g["x"]).mean(), g["kxx_mean"], atol=1e-6)
    close(onp.gaussian_kernel(g["y"], g["y"]).mean(), g["kyy_mean"], atol=1e-6)
    close(onp.gaussian_kernel(g["x"], g["y"]).mean(), g["kxy_mean"], atol=1e-6)
    # the statistic is a difference of O(1) means: absolute tolerance against the component size
    close(onp.maximum_mean_discrepancy(g["y"], g["x"]), g["mmd"], rtol=0, atol=5e-7)


def test_tts_mmd_self_consistency():
    """torch-two-sample is absent: PARITY UNPINNED.  Known-answer checks of the restated formula."""
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((40, 8)), rng.standard_normal((50, 8)) + 0.5
    alphas = [0.1 * i for i in range(15)]
    mmd, K = onp.mmd_statistic_tts(a, b, alphas, ret_matrix=True)
    assert K.shape == (90, 90) and np.allclose(K, K.T)
    # alpha = 0 contributes a constant 1 to every entry and exactly 0 to the statistic
    close(mmd, onp.mmd_statistic_tts(a, b, alphas[1:]), rtol=0, atol=1e-12)
    # identity permutation reproduces the statistic
    lab = np.r_[np.zeros(40, np.uint8), np.ones(50, np.uint8)]
    close(onp.mmd_permutation_stats(K, 40, 50, lab[None])[0], mmd, rtol=0, atol=1e-12)
    assert mmd > 0.01
    same = onp.mmd_statistic_tts(a, rng.standard_normal((50, 8)), alphas)
    assert abs(same) < mmd / 5


# ---- the plain-C oracle is pinned to the same golden vectors ------------------------------------
@pytest.mark.parametrize("case", LOGQZ_CASES)
def test_c_oracle_log_qz(case):
    import math
    from oracle import oracle_c
    g = load_golden("logqz_" + case)
    tol = dict(rtol=1e-4, atol=2e-3) if case in ("peaked", "far") else dict(rtol=2e-5, atol=1e-4)
    lq, pm = oracle_c.log_qz(g["z"], g["mu"], g["logvar"], math.log(g["mu"].shape[0]))
    close(lq, g["log_q_z_all"], **tol)
    close(pm, g["log_q_z_prod_marg_all"], **tol)


@pytest.mark.parametrize("case", ["k1", "k600", "k37_d64"])
def test_c_oracle_iw(case):
    from oracle import oracle_c
    g = load_golden("iw_" + case)
    z, lq, lp, iw = oracle_c.reparam_iw(g["mu"], g["logvar"], g["eps"], g["log_p_x_z"])
    close(z, g["z"], rtol=1e-6, atol=1e-6)
    close(lq, g["log_q_z_x"], atol=1e-4)
    close(lp, g["log_p_z"], atol=1e-4)
    close(iw, g["iw_ll"], atol=1e-4)


@pytest.mark.parametrize("case", ["generic", "ragged", "d64"])
def test_torch_eager_baseline(case):
    import math
    import torch
    from oracle import torch_eager
    g = load_golden("logqz_" + case)
    lq, pm = torch_eager.log_qz(torch.from_numpy(g["mu"]), torch.from_numpy(g["logvar"]),
                                torch.from_numpy(g["z"]), math.log(g["mu"].shape[0]), True, row_chunk=50)
    close(lq.numpy(), g["log_q_z_all"], rtol=1e-5, atol=1e-4)
    close(pm.numpy(), g["log_q_z_prod_marg_all"], rtol=1e-5, atol=1e-4)


def test_kde_batch_is_normalised_by_the_fit_size():
    g = load_golden("kde_batch")
    bw = float(g["bw"])
    close(bw, onp.scott_bandwidth(512, 1), rtol=1e-12, atol=0)
    close(onp.kde_score_samples(g["X"], g["Y"], bw), g["score_full"], atol=1e-5)
    close(onp.kde_score_samples(g["X"][:128], g["Y"], bw, n_fit=512), g["score_batch_128"], atol=1e-5)
    close(onp.kde_score_samples(g["X"][300:], g["Y"], bw, n_fit=512), g["score_batch_tail"], atol=1e-5)
    close(onp.kde_score_samples(g["X2"][:77], g["Y2"], 0.4, n_fit=300), g["score2_batch_77"], atol=1e-5)


def test_gaussian_log_likelihood_generic_masks_and_axes():
    g = load_golden("gll_generic")
    f = onp.gaussian_log_likelihood_masked
    z3 = g["z"][:, None, :]
    close(f(z3, g["mask_sn"], 2, g["mu"], g["var"]), g["a_full_dim2"], atol=2e-4)
    close(f(z3, g["mask_sn"], 2, None, g["var"]), g["a_nomu_dim2"], atol=2e-4)
    close(f(z3, g["mask_sn"], 2, g["mu"], None), g["a_novar_dim2"], atol=2e-4)
    close(f(z3, g["mask_sn"][:, :1], 2), g["a_std_dim2"], atol=2e-4)
    close(f(z3, g["mask_sd"], 1, g["mu"], g["var"]), g["b_full_dim1"], atol=5e-4)
    close(f(z3, g["mask_nd"], 0, g["mu"], g["var"]), g["c_full_dim0"], atol=5e-4)
    close(f(g["s2"], g["mask_b"], 1, g["mu2"], g["var2"]), g["d_full_dim1"], atol=2e-4)
    close(f(g["s2"], g["mask_b"], 1), g["d_std_dim1"], atol=2e-4)
    close(f(g["s2"], g["mask_b"], -1, g["mu2"], None), g["d_novar_dimm1"], atol=2e-4)


def test_c_oracle_masked_logsumexp_reproduces_the_golden_upper_bound():
    """oracle_c.log_qz_excl (used by the K = 20,000 GPU test) against the reference's MBU bound (quirk Q3:
    variance = std, i.e. "logvar" = 0.5 * logvar)."""
    from oracle import oracle_c
    import math
    g = load_golden("mi_bounds_k128")
    z, mulv = g["z"], g["mu_logvars"]
    D = mulv.shape[1] // 2
    mu, lv = mulv[:, :D], mulv[:, D:]
    K = len(z)
    diag = onp.sample_log_likelihood(z, mu, 0.5 * lv, reduce_batch_dim=False)
    off = oracle_c.log_qz_excl(z, mu, 0.5 * lv, 0.0, 0)
    allj, _ = oracle_c.log_qz(z, mu, 0.5 * lv, 0.0, False)
    close(np.mean(diag - (off - math.log(K - 1))), g["upper"], rtol=1e-5, atol=1e-5)
    close(np.mean(diag - (allj - math.log(K))), g["lower"], rtol=1e-5, atol=1e-5)
