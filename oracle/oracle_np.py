"""CPU oracle (NumPy, float64) for the latent-analysis hot path.

TEST INFRASTRUCTURE ONLY.  This module restates, formula by formula, what the reference's Python
functions compute, so that the HIP path can be checked without the reference being present (it never
travels to the GPU box).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package never does and fails loudly when the HIP
library is missing.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against golden vectors
produced by running the reference's own code (``tests/golden/make_golden.py``).  The one exception is
``mmd_statistic_tts`` (torch-two-sample's ``MMDStatistic``): that third-party source is not in
/root/reference (un-vendored submodule josipd/torch-two-sample, no pinned revision, .gitmodules:4-6),
so its restatement follows the published algorithm and is **parity unpinned**.

All citations are path:line under /root/reference/NewsVAE/.  Inputs may be float32; arithmetic is
float64 throughout (the reference computes in float32, so golden comparisons use ~1e-5 tolerances).
"""
import math

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)


def _f64(x):
    return None if x is None else np.asarray(x, dtype=np.float64)


def _logsumexp(a, axis):
    a = np.asarray(a, dtype=np.float64)
    m = np.max(a, axis=axis, keepdims=True)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        out = np.log(np.sum(np.exp(a - m_safe), axis=axis, keepdims=True)) + m_safe
    return np.squeeze(out, axis=axis)


# ----------------------------------------------------------------------------------------------
# loss_and_optimisation.py
# ----------------------------------------------------------------------------------------------
def kl_divergence(mu, logvar, hinge_kl_loss_lambda=0.5, average_batch=True, sum_latent=True):
    """loss_and_optimisation.py:14-43 -- analytic KL and its free-bits (hinge) variant."""
    mu, logvar = _f64(mu), _f64(logvar)
    kl = 0.5 * (mu ** 2 + np.exp(logvar) - logvar - 1.0)
    if hinge_kl_loss_lambda > 0.0:
        hinge = np.where(kl > hinge_kl_loss_lambda, kl, 0.0)      # :27-30 float mask * kl
    else:
        hinge = kl
    if sum_latent:
        hinge, kl = hinge.sum(axis=1), kl.sum(axis=1)
    if average_batch:
        hinge, kl = hinge.mean(axis=0), kl.mean(axis=0)
    return kl, hinge


def gaussian_kernel(x, y):
    """loss_and_optimisation.py:63-76 -- exp(-mean_d((x-y)^2)/D) for every (x_i, y_j)."""
    x, y = _f64(x), _f64(y)
    dim = x.shape[1]
    d2 = ((x[:, None, :] - y[None, :, :]) ** 2).mean(axis=2) / float(dim)
    return np.exp(-d2)


def maximum_mean_discrepancy(latent_z, x):
    """loss_and_optimisation.py:46-60 with the prior draw ``x`` injected (the reference draws
    ``randn_like(latent_z)`` inside, :53).  Biased estimate: diagonals included."""
    y = latent_z
    return gaussian_kernel(x, x).mean() + gaussian_kernel(y, y).mean() - 2.0 * gaussian_kernel(x, y).mean()


def sample_log_likelihood(latent_z, mu=None, logvar=None, reduce_latent_dim=True, reduce_batch_dim=True):
    """loss_and_optimisation.py:79-110."""
    z, mu, logvar = _f64(latent_z), _f64(mu), _f64(logvar)
    if z.ndim == 3 and mu is not None and logvar is not None:      # :86-90
        if mu.ndim != 3:
            mu = mu[:, None, :]
        if logvar.ndim != 3:
            logvar = logvar[:, None, :]
    if logvar is not None and mu is not None:                      # :93-96
        ll = -0.5 * (logvar + LOG_2PI + (z - mu) ** 2 / np.exp(logvar))
    else:                                                          # :99-101
        ll = -0.5 * (LOG_2PI + z ** 2)
    if reduce_latent_dim:
        ll = ll.sum(axis=-1)
    if reduce_batch_dim:
        ll = ll.mean(axis=0)
    return ll


def approximate_log_q_z(mu, logvar, latent_z, method="chen", dataset_size=42068, prod_marginals=False,
                        reduce_mean=True, row_chunk=256):
    """loss_and_optimisation.py:139-221.  Row-chunked over z (exact: both logsumexps run over the
    posterior axis only, :195,:198).  Quirk Q1 (:219, ``None.mean()``) is reproduced as AttributeError."""
    mu, logvar, z = _f64(mu), _f64(logvar), _f64(latent_z)
    if method == "chen":
        assert dataset_size is not None
    x_batch = mu.shape[0]
    log_norm = math.log(x_batch * dataset_size) if method == "chen" else math.log(x_batch)
    M = z.shape[0]
    log_q_z = np.empty(M)
    log_q_z_pm = np.empty(M) if prod_marginals else None
    inv_var = np.exp(-logvar)
    for r0 in range(0, M, row_chunk):
        zc = z[r0:r0 + row_chunk, None, :]
        log_dens = -0.5 * (logvar[None] + LOG_2PI + (zc - mu[None]) ** 2 * inv_var[None])   # :185
        log_q_z[r0:r0 + row_chunk] = _logsumexp(log_dens.sum(axis=2), axis=1) - log_norm   # :198,207,215
        if prod_marginals:
            log_q_z_pm[r0:r0 + row_chunk] = (_logsumexp(log_dens, axis=1) - log_norm).sum(axis=1)  # :195,205,213
    if reduce_mean:
        if log_q_z_pm is None:
            raise AttributeError("'NoneType' object has no attribute 'mean'")   # Q1, :219
        return log_q_z.mean(), log_q_z_pm.mean()
    return log_q_z, log_q_z_pm


def approximate_marginal_KL_intended(mu, logvar, latent_z, method="chen", dataset_size=None):
    """What loss_and_optimisation.py:113-121 intends: mean_i(log q(z_i) - log p(z_i)).
    The reference itself always raises here (Q1)."""
    log_q_z, _ = approximate_log_q_z(mu, logvar, latent_z, method=method, dataset_size=dataset_size,
                                     prod_marginals=False, reduce_mean=False)
    log_p_z = sample_log_likelihood(latent_z, reduce_latent_dim=True, reduce_batch_dim=False)
    return (log_q_z - log_p_z).mean()


def approximate_total_correlation(mu, logvar, latent_z, method="chen", dataset_size=None):
    """loss_and_optimisation.py:124-136."""
    a, b = approximate_log_q_z(mu, logvar, latent_z, method=method, dataset_size=dataset_size,
                               prod_marginals=True)
    return a - b


def iw_log_p_x(log_p_x_z, log_p_z, log_q_z_x):
    """loss_and_optimisation.py:783-794 (twin: utils_evaluation.py:176-180)."""
    frac = _f64(log_p_x_z) + _f64(log_p_z) - _f64(log_q_z_x)
    return _logsumexp(frac, axis=-1) - np.log(frac.shape[1])


# ----------------------------------------------------------------------------------------------
# modules/encoder.py
# ----------------------------------------------------------------------------------------------
def reparameterize(mu, logvar, eps):
    """modules/encoder.py:130-159 with the noise ``eps`` [B,K,D] injected (drawn inside at :156).
    Computed in float32 like the reference (separate multiply and add) so z matches bit-for-bit
    up to the exp() implementation."""
    mu32, lv32, eps32 = (np.asarray(a, dtype=np.float32) for a in (mu, logvar, eps))
    std = np.exp(np.float32(0.5) * lv32).astype(np.float32)
    return (mu32[:, None, :] + (eps32 * std[:, None, :]).astype(np.float32)).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# mutual_information.py
# ----------------------------------------------------------------------------------------------
def gaussian_log_likelihood(sample, mu=None, var=None):
    """mutual_information.py:26-41 with sample [S,1,D] against [N,D] parameters, mask == 1, dim == 2.
    ``sample`` here is [S,D]; returns [S,N] (or [S,1] for the parameter-free branches)."""
    s = _f64(sample)[:, None, :]
    mu, var = _f64(mu), _f64(var)
    if mu is None and var is None:
        return -0.5 * np.sum(LOG_2PI + s ** 2, axis=2)
    if mu is None:
        return -0.5 * np.sum(np.log(2 * np.pi * var)[None] + s ** 2 / var[None], axis=2)
    if var is None:
        return -0.5 * np.sum(LOG_2PI + (s - mu[None]) ** 2, axis=2)
    return -0.5 * np.sum(np.log(2 * np.pi * var)[None] + (s - mu[None]) ** 2 / var[None], axis=2)


def gaussian_log_likelihood_masked(sample, mask, dim, mu=None, var=None):
    """mutual_information.py:26-41 in full generality: the element-wise term of the branch selected by which of
    mu / var is given, times ``mask`` with a singleton axis inserted at ``dim``, summed over ``dim``
    (NumPy broadcasting == torch broadcasting)."""
    s, m = _f64(sample), _f64(mask)
    if mu is None and var is None:
        t = LOG_2PI + s ** 2                                                # :34
    elif mu is None:
        t = np.log(2 * math.pi * _f64(var)) + s ** 2 / _f64(var)            # :36
    elif var is None:
        t = LOG_2PI + (s - _f64(mu)) ** 2                                   # :38
    else:
        t = np.log(2 * math.pi * _f64(var)) + (s - _f64(mu)) ** 2 / _f64(var)   # :41
    me = np.expand_dims(m, dim if dim >= 0 else m.ndim + 1 + dim)
    return -0.5 * (t * me).sum(axis=dim)


def q_z_estimate(z, mu, var):
    """mutual_information.py:44-52.  Quirk Q2: the logsumexp runs over dim=1 of [1,S,N], i.e. the
    SAMPLE axis, so the result is [1,N]: LSE_i log q(z_i|x_j) - log N per posterior j."""
    m = gaussian_log_likelihood(z, mu=mu, var=var)                  # [S,N]
    return (_logsumexp(m, axis=0) - math.log(m.shape[1]))[None, :]


def calc_mi_estimate(log_q_z_x, log_p_z, avg_KL, mi_method="zhao"):
    """mutual_information.py:56-105, aggregate_posterior branch (:84-85): log_q_z_x is the [N] vector."""
    lq = _f64(log_q_z_x)
    log_q_z = _logsumexp(lq, axis=0) - math.log(len(lq))            # :85
    marg_kl = log_q_z - _f64(log_p_z).mean()                        # :94
    if mi_method == "zhao":
        mi = lq.mean() - log_q_z                                    # :98
    elif mi_method == "hoffman":
        mi = float(avg_KL) - marg_kl                                # :101
    else:
        raise ValueError(mi_method)
    return float(mi), float(marg_kl)


def log_prob_pairs(z, mu, cov_diag):
    """mutual_information.py:108-132 for K diagonal MultivariateNormals: P[j,i] = log N(z_i; mu_j,
    diag(cov_diag_j)).  Rows = distributions, columns = samples."""
    z, mu, c = _f64(z), _f64(mu), _f64(cov_diag)
    diff = z[None, :, :] - mu[:, None, :]
    return -0.5 * np.sum(np.log(2 * np.pi * c)[:, None, :] + diff ** 2 / c[:, None, :], axis=2)


def mi_bounds_from_pairs(P):
    """mutual_information.py:135-215 -- InfoNCE lower and MBU upper bound from the K x K matrix."""
    P = _f64(P)
    K = P.shape[0]
    diag = np.diag(P)
    lower = (diag - (_logsumexp(P, axis=0) - math.log(K))).mean()              # :165-171
    off = P.copy()
    off[np.arange(K), np.arange(K)] = -np.inf                                   # :207
    upper = (diag - (_logsumexp(off, axis=0) - math.log(K - 1))).mean()       # :208-213
    return float(lower), float(upper)


def calc_upper_and_lower_bound_representational_mi(batch_latent_samples, batch_mu_logvars):
    """mutual_information.py:218-248.  Quirk Q3: the distributions are MultivariateNormal(mu,
    diag(std)) (:240-243) -- the second positional argument is the COVARIANCE, so the variance used
    is exp(0.5*logvar), not exp(logvar)."""
    mulv = _f64(batch_mu_logvars)
    D = mulv.shape[1] // 2
    mu, logvar = mulv[:, :D], mulv[:, D:]
    return mi_bounds_from_pairs(log_prob_pairs(batch_latent_samples, mu, np.exp(0.5 * logvar)))


def categorical_log_prob_pairs(probs, samples):
    """log_prob_pairs (mutual_information.py:108-132) on torch Categorical(probs=row) critics, as
    calc_upper_and_lower_bound_generative_mi builds them (:271-273): torch normalises the row and takes
    log(clamp(p, eps, 1 - eps)) with the fp32 machine epsilon (torch.distributions.utils.probs_to_logits),
    then gathers the sampled class.  P[j, i] = log p_j(samples[i])."""
    p = np.asarray(probs, dtype=np.float32)
    pn = p / p.sum(axis=-1, keepdims=True, dtype=np.float32)
    eps = np.finfo(np.float32).eps
    logits = np.log(np.clip(pn, eps, 1 - eps).astype(np.float64))
    return logits[:, np.asarray(samples, dtype=np.int64)]


def calc_upper_and_lower_bound_generative_mi(batch_probs, sampled_predictions):
    """mutual_information.py:251-278 with the multinomial draw (:269) injected.  As written the function
    only broadcasts for seq_len == 1 (diag [K] against logsumexp [K * seq_len], :165-171), which is the
    case restated here: batch_probs [K, 1, V] or [K, V]."""
    p = np.asarray(batch_probs)
    p = p.reshape(-1, p.shape[-1])
    return mi_bounds_from_pairs(categorical_log_prob_pairs(p, sampled_predictions))


# ----------------------------------------------------------------------------------------------
# utils_train.py: Pareto-efficient checkpoints and the KS distance between IW-LL vectors
# ----------------------------------------------------------------------------------------------
def ks_2samp_statistic(a, b):
    """The two-sided statistic of scipy.stats.ks_2samp as called at utils_train.py:496-497 (third-party:
    scipy, 1.15.3 in this image): both samples sorted, empirical CDFs evaluated at every pooled value with
    side='right', D = max(max(F_a - F_b), clip(-min(F_a - F_b), 0, 1))."""
    a, b = np.sort(_f64(a).ravel()), np.sort(_f64(b).ravel())
    pooled = np.concatenate([a, b])
    d = np.searchsorted(a, pooled, side="right") / a.size - np.searchsorted(b, pooled, side="right") / b.size
    return float(max(d.max(), min(max(-d.min(), 0.0), 1.0)))


def is_pareto_efficient(gains):
    """utils_train.py:416-431 (gains: higher is better): a point is dropped iff another point is >= it in
    every coordinate; of several identical points the first one is kept (:429-430).  Brute force O(n^2)."""
    g = _f64(gains)
    n = g.shape[0]
    keep = np.ones(n, dtype=bool)
    for i in range(n):
        for j in range(n):
            if j != i and np.all(g[i] <= g[j]) and (np.any(g[i] < g[j]) or j < i):
                keep[i] = False
    return keep


# ----------------------------------------------------------------------------------------------
# utils_latent_analysis_optimisation.py
# ----------------------------------------------------------------------------------------------
def scott_bandwidth(n, d):
    """utils_latent_analysis_optimisation.py:24-26."""
    return n ** (-1.0 / (d + 4))


def kde_score_samples(X, Y, bw, n_fit=None):
    """utils_latent_analysis_optimisation.py:55-58.  Quirk Q6: exp -> sum -> log with no
    stabilisation, so a query far from every point scores -inf.  To mirror the reference's float32
    underflow the kernel values are flushed where a float32 exp would underflow to zero.
    ``n_fit``: the reference divides by self.n, the size of the set the estimator was CONSTRUCTED on, also
    when score_samples is handed a batch X of another size (:58)."""
    X, Y = _f64(X), _f64(Y)
    n, d = X.shape
    if n_fit is not None:
        n = n_fit
    u = (X[:, None, :] - Y[None, :, :]) / bw                        # [n,m,d]
    logk = -0.5 * (u ** 2).sum(axis=2) - 0.5 * d * LOG_2PI
    k = np.exp(logk)
    k[logk < math.log(2.0 ** -149)] = 0.0
    with np.errstate(divide="ignore"):
        return np.log(bw ** (-d) * k.sum(axis=0) / n)


def get_1d_kde_dim_marg_kl_estimate(latents, log_p_z, bw="scott"):
    """loss_and_optimisation.py:625-637."""
    lat = _f64(latents)
    flat = lat.reshape(-1, 1)
    if bw == "scott":
        bw = scott_bandwidth(flat.shape[0], 1)
    s = kde_score_samples(flat, flat, bw).reshape(lat.shape)
    return s.sum(axis=1).mean(axis=0) - _f64(log_p_z)


# ----------------------------------------------------------------------------------------------
# torch_two_sample.statistics_diff.MMDStatistic  (EXTERNAL, ABSENT -> parity unpinned)
# ----------------------------------------------------------------------------------------------
def mmd_statistic_tts(sample_1, sample_2, alphas, ret_matrix=False):
    """josipd/torch-two-sample ``MMDStatistic(n_1, n_2)(sample_1, sample_2, alphas)`` as called at
    loss_and_optimisation.py:617-619.  PARITY UNPINNED: restated from the published algorithm:
    pooled distances d = sqrt(1e-5 + |a.a + b.b - 2 a.b|); K = sum_alpha exp(-alpha d^2);
    mmd = 2 a01 sum(K12) + a00 (sum(K11) - tr K11) + a11 (sum(K22) - tr K22)."""
    s1, s2 = _f64(sample_1), _f64(sample_2)
    n1, n2 = s1.shape[0], s2.shape[0]
    s12 = np.concatenate([s1, s2], axis=0)
    nrm = (s12 ** 2).sum(axis=1)
    d2 = 1e-5 + np.abs(nrm[:, None] + nrm[None, :] - 2.0 * (s12 @ s12.T))
    K = np.zeros_like(d2)
    for a in alphas:
        K += np.exp(-a * d2)
    a00, a11, a01 = 1.0 / (n1 * (n1 - 1)), 1.0 / (n2 * (n2 - 1)), -1.0 / (n1 * n2)
    k1, k2, k12 = K[:n1, :n1], K[n1:, n1:], K[:n1, n1:]
    mmd = 2 * a01 * k12.sum() + a00 * (k1.sum() - np.trace(k1)) + a11 * (k2.sum() - np.trace(k2))
    return (mmd, K) if ret_matrix else mmd


def mmd_permutation_stats(K, n1, n2, perms):
    """Statistic of ``mmd_statistic_tts`` re-evaluated under label permutations.  ``perms`` is
    [P, n1+n2] uint8 with exactly n1 zeros per row (0 -> sample 1, 1 -> sample 2)."""
    a00, a11, a01 = 1.0 / (n1 * (n1 - 1)), 1.0 / (n2 * (n2 - 1)), -1.0 / (n1 * n2)
    out = np.empty(len(perms))
    dg = np.diag(K)
    for p, lab in enumerate(perms):
        s = lab.astype(np.float64)
        u = 1.0 - s
        Ku, Ks = K @ u, K @ s
        out[p] = (a00 * (u @ Ku - (dg * u).sum()) + a11 * (s @ Ks - (dg * s).sum())
                  + 2 * a01 * (u @ Ks))
    return out
