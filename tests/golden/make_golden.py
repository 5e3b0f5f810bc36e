#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the REFERENCE's own
functions (imported from /root/reference/NewsVAE) on seeded synthetic inputs.

Runs only in the build container (the reference never travels to the GPU box); it is a no-op when
/root/reference is absent.  The committed ``*.npz`` files are data only: inputs and the outputs the
reference produced for them.  Nothing from this script is imported by the product code.

Third-party modules the reference imports at module scope but never touches on the hot path
(``constraintoptim``, ``torch_two_sample``, the training drivers) are registered as inert stubs so
that the import succeeds; see SURVEY.md section 8(c).  ``torch_two_sample`` itself is absent, so
``get_tts_mmmd`` has NO golden vector ("parity unpinned").

Usage:  python tests/golden/make_golden.py
"""
import math
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference/NewsVAE"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    class _Inert:  # placeholder class for names imported at module scope
        def __init__(self, *a, **k):
            raise RuntimeError("inert stub")

    _stub("constraintoptim")
    _stub("constraintoptim.constraint", Constraint=_Inert, ConstraintOptimizer=_Inert)
    tts = _stub("torch_two_sample")
    tts.statistics_diff = _stub("torch_two_sample.statistics_diff", MMDStatistic=_Inert)
    _stub("train", get_model_on_device=None)
    _stub("utils_evaluation", valid_dataset_loader_tokenizer=None)
    _stub("utils_train", load_from_checkpoint=None, transfer_batch_to_device=None)
    _stub("modules.encoder_roberta", VAE_Encoder_RobertaModel=_Inert)

    import loss_and_optimisation as lo
    import utils_latent_analysis_optimisation as kde
    import modules.encoder as enc
    import mutual_information as mi
    return lo, kde, enc, mi


def gen_inputs(seed, N, M, D, mode="generic"):
    """Synthetic posteriors and samples, SURVEY.md section 8(d) distributions."""
    g = torch.Generator().manual_seed(seed)
    if mode == "collapsed":
        mu = torch.zeros(N, D)
        logvar = torch.zeros(N, D)
    elif mode == "near_collapsed":
        mu = 0.1 * torch.randn(N, D, generator=g)
        logvar = 0.1 * torch.randn(N, D, generator=g)
    elif mode == "peaked":
        mu = torch.randn(N, D, generator=g)
        logvar = 0.3 * torch.randn(N, D, generator=g) - 10.0
    else:
        mu = torch.randn(N, D, generator=g)
        logvar = 0.5 * torch.randn(N, D, generator=g) - 1.0
    eps = torch.randn(M, D, generator=g)
    idx = torch.arange(M) % N
    z = mu[idx] + eps * torch.exp(0.5 * logvar[idx])
    if mode == "far":
        z = z + 25.0 * torch.sign(torch.randn(M, D, generator=g))
    return mu, logvar, z


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def np_any(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def case_logqz(lo, name, seed, N, M, D, mode):
    mu, logvar, z = gen_inputs(seed, N, M, D, mode)
    out = {"mu": np32(mu), "logvar": np32(logvar), "z": np32(z)}
    for tag, method, ds in (("chen10000", "chen", 10000), ("chen42068", "chen", 42068),
                            ("all", "all", None)):
        lq, lqpm = lo.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds,
                                          prod_marginals=True, reduce_mean=False)
        out[f"log_q_z_{tag}"] = np32(lq)
        out[f"log_q_z_prod_marg_{tag}"] = np32(lqpm)
        lqm, lqpmm = lo.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds,
                                            prod_marginals=True, reduce_mean=True)
        out[f"log_q_z_mean_{tag}"] = np32(lqm)
        out[f"log_q_z_prod_marg_mean_{tag}"] = np32(lqpmm)
        lq2, none = lo.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds,
                                           prod_marginals=False, reduce_mean=False)
        assert none is None
        out[f"log_q_z_noprod_{tag}"] = np32(lq2)
    out["total_correlation_chen42068"] = np32(
        lo.approximate_total_correlation(mu, logvar, z, method="chen", dataset_size=42068))
    # Q1: approximate_marginal_KL raises as written (None.mean()); record that it did
    try:
        lo.approximate_marginal_KL(mu, logvar, z, method="chen", dataset_size=42068)
        out["marginal_kl_raises"] = np.array(0)
    except AttributeError:
        out["marginal_kl_raises"] = np.array(1)
    # per-sample densities (M == N only: z_i belongs to posterior i)
    out["log_p_z"] = np32(lo.sample_log_likelihood(z, reduce_latent_dim=True, reduce_batch_dim=False))
    out["log_p_z_nored"] = np32(lo.sample_log_likelihood(z, reduce_latent_dim=False,
                                                         reduce_batch_dim=False))
    out["log_p_z_batchmean"] = np32(lo.sample_log_likelihood(z))
    if M == N:
        out["log_q_z_x"] = np32(lo.sample_log_likelihood(z, mu, logvar, reduce_latent_dim=True,
                                                         reduce_batch_dim=False))
        out["log_q_z_x_nored"] = np32(lo.sample_log_likelihood(z, mu, logvar,
                                                               reduce_latent_dim=False,
                                                               reduce_batch_dim=False))
    for lam in (0.0, 0.5):
        kl, hk = lo.kl_divergence(mu, logvar, hinge_kl_loss_lambda=lam)
        out[f"kl_lam{lam}"] = np32(kl)
        out[f"hinge_kl_lam{lam}"] = np32(hk)
        kl, hk = lo.kl_divergence(mu, logvar, hinge_kl_loss_lambda=lam, average_batch=False,
                                  sum_latent=True)
        out[f"kl_rows_lam{lam}"] = np32(kl)
        out[f"hinge_kl_rows_lam{lam}"] = np32(hk)
        kl, hk = lo.kl_divergence(mu, logvar, hinge_kl_loss_lambda=lam, average_batch=False,
                                  sum_latent=False)
        out[f"kl_elem_lam{lam}"] = np32(kl)
        out[f"hinge_kl_elem_lam{lam}"] = np32(hk)
    np.savez_compressed(os.path.join(OUT, f"logqz_{name}.npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if v.ndim > 0 and k.startswith("log_q_z_a")})


def case_mi(lo, mi, name, seed, K, D):
    mu, logvar, z = gen_inputs(seed, K, K, D, "generic")
    mulv = torch.cat([mu, logvar], dim=1)
    lower, upper = mi.calc_upper_and_lower_bound_representational_mi(z, mulv)
    std = torch.exp(0.5 * logvar)
    dists = [torch.distributions.MultivariateNormal(mu[i], torch.diag(std[i])) for i in range(K)]
    pairs = mi.log_prob_pairs(dists, z)
    out = {"z": np32(z), "mu_logvars": np32(mulv), "lower": np.float64(lower),
           "upper": np.float64(upper), "log_prob_pairs": np32(pairs)}
    np.savez_compressed(os.path.join(OUT, f"mi_bounds_{name}.npz"), **out)
    print("wrote mi", name, lower, upper)


def case_qz_estimate(mi, seed, S, N, D):
    import contextlib
    import io
    mu, logvar, z = gen_inputs(seed, N, S, D, "generic")
    var = logvar.exp()
    with contextlib.redirect_stdout(io.StringIO()):
        q = mi.q_z_estimate(z, mu, var, device="cpu")
    one = torch.tensor([[1.]])
    out = {"z": np32(z), "mu": np32(mu), "var": np32(var), "q_z_estimate": np32(q),
           "gll_full": np32(mi._gaussian_log_likelihood(z.unsqueeze(1), one, 2, mu=mu, var=var)),
           "gll_nomu": np32(mi._gaussian_log_likelihood(z.unsqueeze(1), one, 2, mu=None, var=var)),
           "gll_novar": np32(mi._gaussian_log_likelihood(z.unsqueeze(1), one, 2, mu=mu, var=None)),
           "gll_std": np32(mi._gaussian_log_likelihood(z.unsqueeze(1), one, 2))}
    # calc_mi_estimate on the diagonal vector (N == S rows only)
    n = min(S, N)
    lqzx = out["gll_full"][np.arange(n), np.arange(n)]
    lqzx_t = torch.from_numpy(lqzx)
    log_p_z = (-0.5 * (math.log(2 * math.pi) + z[:n] ** 2)).sum(-1)
    avg_kl = (0.5 * (mu[:n] ** 2 + logvar[:n].exp() - logvar[:n] - 1)).sum(1).mean()
    with contextlib.redirect_stdout(io.StringIO()):
        mi_z, mkl_z = mi.calc_mi_estimate(lqzx_t, z[:n], mu[:n], logvar[:n], log_p_z, avg_kl,
                                          mi_method="zhao")
        mi_h, mkl_h = mi.calc_mi_estimate(lqzx_t, z[:n], mu[:n], logvar[:n], log_p_z, avg_kl,
                                          mi_method="hoffman")
    out.update(cme_log_q_z_x=lqzx, cme_log_p_z=np32(log_p_z), cme_avg_kl=np32(avg_kl),
               cme_mi_zhao=np.float64(mi_z), cme_marg_kl_zhao=np.float64(mkl_z),
               cme_mi_hoffman=np.float64(mi_h), cme_marg_kl_hoffman=np.float64(mkl_h))
    np.savez_compressed(os.path.join(OUT, "qz_estimate.npz"), **out)
    print("wrote qz_estimate", q.shape)


def case_iw(lo, enc, name, seed, B, K, D):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(B, D, generator=g)
    logvar = 0.5 * torch.randn(B, D, generator=g) - 1.0
    log_p_x_z = 5.0 * torch.randn(B, K, generator=g) - 100.0
    torch.manual_seed(seed + 1000)
    z = enc.EncoderNewsVAE.reparameterize(mu, logvar, n_samples=K)
    torch.manual_seed(seed + 1000)
    eps = torch.zeros(B, K, D).normal_()
    z_chk = mu.unsqueeze(1) + eps * torch.exp(0.5 * logvar).unsqueeze(1)
    assert torch.equal(z, z_chk), "eps draw order differs from a contiguous [B,K,D] normal_()"
    log_q_z_x = lo.sample_log_likelihood(z, mu=mu, logvar=logvar, reduce_latent_dim=True,
                                         reduce_batch_dim=False)
    log_p_z = lo.sample_log_likelihood(z, mu=None, logvar=None, reduce_latent_dim=True,
                                       reduce_batch_dim=False)
    iw = lo.LossTermManager.iw_log_p_x(None, log_p_x_z, log_p_z, log_q_z_x)
    out = {"mu": np32(mu), "logvar": np32(logvar), "eps": np32(eps), "log_p_x_z": np32(log_p_x_z),
           "z": np32(z), "log_q_z_x": np32(log_q_z_x), "log_p_z": np32(log_p_z), "iw_ll": np32(iw)}
    np.savez_compressed(os.path.join(OUT, f"iw_{name}.npz"), **out)
    print("wrote iw", name, iw.shape)


def case_kde(lo, kde, seed):
    g = torch.Generator().manual_seed(seed)
    B, D = 16, 32
    lat = torch.randn(B, D, generator=g) * 1.3 + 0.2
    out = {"latents": np32(lat)}
    k = kde.GaussianKDE(X=lat.reshape(-1, 1), bw="scott", device="cpu")
    out["bw_scott_1d"] = np.float64(k.bw)
    out["score_1d_scott"] = np32(k.score_samples(lat.reshape(-1, 1)))
    k2 = kde.GaussianKDE(X=lat.reshape(-1, 1), bw=0.25, device="cpu")
    Y = torch.cat([lat.reshape(-1, 1)[:50], torch.tensor([[40.0], [-60.0], [1e3]])], 0)
    out["Y_1d"] = np32(Y)
    out["score_1d_bw025"] = np32(k2.score_samples(Y))  # last rows underflow to -inf (Q6)
    X3 = torch.randn(200, 3, generator=g)
    Y3 = torch.randn(64, 3, generator=g)
    k3 = kde.GaussianKDE(X=X3, bw="scott", device="cpu")
    out.update(X3=np32(X3), Y3=np32(Y3), bw_scott_3d=np.float64(k3.bw),
               score_3d_scott=np32(k3.score_samples(Y3)))
    log_p_z = lo.sample_log_likelihood(lat, reduce_latent_dim=True, reduce_batch_dim=True)
    out["log_p_z_mean"] = np32(log_p_z)
    out["kde1d_dim_marg_kl"] = np32(lo.LossTermManager.get_1d_kde_dim_marg_kl_estimate(
        lat, log_p_z, bw="scott", device="cpu"))
    np.savez_compressed(os.path.join(OUT, "kde.npz"), **out)
    print("wrote kde")


def case_kde_batch(kde, seed):
    """score_samples(Y, X=batch): the reference still divides by self.n, the size of the FIT set
    (utils_latent_analysis_optimisation.py:58)."""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(512, 1, generator=g) * 1.1
    Y = torch.randn(90, 1, generator=g)
    k = kde.GaussianKDE(X=X, bw="scott", device="cpu")
    out = {"X": np32(X), "Y": np32(Y), "bw": np.float64(k.bw),
           "score_full": np32(k.score_samples(Y)),
           "score_batch_128": np32(k.score_samples(Y, X=X[:128])),
           "score_batch_tail": np32(k.score_samples(Y, X=X[300:]))}
    X2 = torch.randn(300, 2, generator=g)
    Y2 = torch.randn(40, 2, generator=g)
    k2 = kde.GaussianKDE(X=X2, bw=0.4, device="cpu")
    out.update(X2=np32(X2), Y2=np32(Y2), score2_batch_77=np32(k2.score_samples(Y2, X=X2[:77])))
    np.savez_compressed(os.path.join(OUT, "kde_batch.npz"), **out)
    print("wrote kde_batch")


def case_gll_generic(mi, seed):
    """_gaussian_log_likelihood (mutual_information.py:26-41) away from the one layout q_z_estimate uses:
    non-trivial masks, other reduction axes, 2-D samples."""
    g = torch.Generator().manual_seed(seed)
    S, N, D = 37, 21, 12
    z = torch.randn(S, D, generator=g)
    mu = torch.randn(N, D, generator=g)
    var = (0.4 * torch.randn(N, D, generator=g)).exp()
    mask_sn = (torch.rand(S, N, generator=g) > 0.3).float()          # one weight per (sample, posterior) pair
    mask_sd = torch.rand(S, D, generator=g)                           # dim=1: weights per (sample, dimension)
    mask_nd = (torch.rand(N, D, generator=g) > 0.5).float()          # dim=0: weights per (posterior, dimension)
    f = mi._gaussian_log_likelihood
    out = {"z": np32(z), "mu": np32(mu), "var": np32(var), "mask_sn": np32(mask_sn), "mask_sd": np32(mask_sd),
           "mask_nd": np32(mask_nd),
           "a_full_dim2": np32(f(z.unsqueeze(1), mask_sn, 2, mu=mu, var=var)),        # [S,N]
           "a_nomu_dim2": np32(f(z.unsqueeze(1), mask_sn, 2, mu=None, var=var)),
           "a_novar_dim2": np32(f(z.unsqueeze(1), mask_sn, 2, mu=mu, var=None)),
           "a_std_dim2": np32(f(z.unsqueeze(1), mask_sn[:, :1], 2)),                  # [S,1]
           "b_full_dim1": np32(f(z.unsqueeze(1), mask_sd, 1, mu=mu, var=var)),        # [S,D]: summed over posteriors
           "c_full_dim0": np32(f(z.unsqueeze(1), mask_nd, 0, mu=mu, var=var))}        # [N,D]: summed over samples
    # 2-D layout of the function's origin (sentence x position): sample, mu, var [B,L], mask [B,L], dim=1... the
    # mask is unsqueezed at `dim`, so a [B] mask with dim=1 weights whole rows
    B, L = 19, 9
    s2 = torch.randn(B, L, generator=g)
    mu2 = torch.randn(B, L, generator=g)
    var2 = (0.3 * torch.randn(B, L, generator=g)).exp()
    mask_b = (torch.rand(B, generator=g) > 0.4).float()
    out.update(s2=np32(s2), mu2=np32(mu2), var2=np32(var2), mask_b=np32(mask_b),
               d_full_dim1=np32(f(s2, mask_b, 1, mu=mu2, var=var2)),                  # [B]
               d_std_dim1=np32(f(s2, mask_b, 1)),
               d_novar_dimm1=np32(f(s2, mask_b, -1, mu=mu2, var=None)))               # dim=-1: mask [B] -> [B,1]
    np.savez_compressed(os.path.join(OUT, "gll_generic.npz"), **out)
    print("wrote gll_generic")


def case_mmd(lo, seed, n, D, name):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(n, D, generator=g)
    y = mu + torch.randn(n, D, generator=g) * 0.6
    torch.manual_seed(seed + 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mmd = lo.maximum_mean_discrepancy(y)
        torch.manual_seed(seed + 7)
        x = torch.randn_like(y)
        kxx, kyy, kxy = lo.gaussian_kernel(x, x), lo.gaussian_kernel(y, y), lo.gaussian_kernel(x, y)
    mmd2 = kxx.mean() + kyy.mean() - 2 * kxy.mean()
    assert torch.equal(mmd, mmd2), "randn_like draw does not reproduce"
    m = min(n, 48)
    out = {"x": np32(x), "y": np32(y), "mmd": np32(mmd), "kxx_mean": np32(kxx.mean()),
           "kyy_mean": np32(kyy.mean()), "kxy_mean": np32(kxy.mean()),
           "kxy_block": np32(kxy[:m, :m])}
    np.savez_compressed(os.path.join(OUT, f"mmd_{name}.npz"), **out)
    print("wrote mmd", name, float(mmd))


def case_grad(lo, name, seed, N, M, D, mode="generic"):
    """Gradients of the reference's approximate_log_q_z under torch autograd (how its training objectives
    use it, loss_and_optimisation.py:925-1018): loss = sum_i a_i log_qz_i + sum_i c_i log_qz_prod_marg_i."""
    mu, logvar, z = gen_inputs(seed, N, M, D, mode)
    g = torch.Generator().manual_seed(seed + 1000)
    a = torch.randn(M, generator=g)
    c = torch.randn(M, generator=g)
    out = {"mu": np32(mu), "logvar": np32(logvar), "z": np32(z), "a": np32(a), "c": np32(c)}
    for tag, wa, wc in (("joint", a, None), ("pm", None, c), ("both", a, c)):
        mu_, lv_, z_ = (t.clone().requires_grad_(True) for t in (mu, logvar, z))
        lq, lqpm = lo.approximate_log_q_z(mu_, lv_, z_, method="chen", dataset_size=10000,
                                          prod_marginals=True, reduce_mean=False)
        loss = 0.0
        if wa is not None:
            loss = loss + (wa * lq).sum()
        if wc is not None:
            loss = loss + (wc * lqpm).sum()
        loss.backward()
        out[f"d_mu_{tag}"], out[f"d_logvar_{tag}"], out[f"d_z_{tag}"] = np32(mu_.grad), np32(lv_.grad), np32(z_.grad)
    # the mean-reduced form the objectives use: TC = mean(log_qz) - mean(log_qz_prod_marg)
    mu_, lv_, z_ = (t.clone().requires_grad_(True) for t in (mu, logvar, z))
    tc = lo.approximate_total_correlation(mu_, lv_, z_, method="chen", dataset_size=10000)
    tc.backward()
    out["tc"] = np32(tc)
    out["d_mu_tc"], out["d_logvar_tc"], out["d_z_tc"] = np32(mu_.grad), np32(lv_.grad), np32(z_.grad)
    np.savez_compressed(os.path.join(OUT, f"grad_{name}.npz"), **out)
    print("wrote grad", name)


def case_grad_elem(lo, seed, B, K, D):
    """Gradients of sample_log_likelihood (posterior / prior, reduced / element-wise) and kl_divergence."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(B, D, generator=g)
    logvar = 0.5 * torch.randn(B, D, generator=g) - 1.0
    z = mu.unsqueeze(1) + torch.randn(B, K, D, generator=g) * torch.exp(0.5 * logvar).unsqueeze(1)
    w_bk = torch.randn(B, K, generator=g)
    w_bkd = torch.randn(B, K, D, generator=g)
    out = {"mu": np32(mu), "logvar": np32(logvar), "z": np32(z), "w_bk": np32(w_bk), "w_bkd": np32(w_bkd)}

    def run(tag, fn):
        mu_, lv_, z_ = (t.clone().requires_grad_(True) for t in (mu, logvar, z))
        val = fn(mu_, lv_, z_)
        val.backward()
        out[f"val_{tag}"] = np32(val)
        for nm, t in (("mu", mu_), ("logvar", lv_), ("z", z_)):
            out[f"d_{nm}_{tag}"] = np32(t.grad) if t.grad is not None else np.zeros(t.shape, np.float32)

    run("post_red", lambda m, l, zz: (w_bk * lo.sample_log_likelihood(zz, m, l, True, False)).sum())
    run("post_elem", lambda m, l, zz: (w_bkd * lo.sample_log_likelihood(zz, m, l, False, False)).sum())
    run("post_mean", lambda m, l, zz: lo.sample_log_likelihood(zz, m, l).sum())
    run("prior_red", lambda m, l, zz: (w_bk * lo.sample_log_likelihood(zz, reduce_batch_dim=False)).sum())
    run("post_2d", lambda m, l, zz: (w_bk[:, 0] * lo.sample_log_likelihood(zz[:, 0], m, l, True, False)).sum())
    for lam in (0.0, 0.5):
        run(f"kl_lam{lam}", lambda m, l, zz: sum(a * b for a, b in zip((1.0, 0.7), lo.kl_divergence(m, l, lam))))
        run(f"kl_rows_lam{lam}", lambda m, l, zz: sum((w_bk[:, 0] * t).sum() * c for t, c in zip(
            lo.kl_divergence(m, l, lam, average_batch=False, sum_latent=True), (1.0, -0.3))))
        run(f"kl_elem_lam{lam}", lambda m, l, zz: sum((w_bkd[:, 0] * t).sum() * c for t, c in zip(
            lo.kl_divergence(m, l, lam, average_batch=False, sum_latent=False), (0.4, 1.0))))
    np.savez_compressed(os.path.join(OUT, "grad_elem.npz"), **out)
    print("wrote grad_elem")


def case_grad_encode_tail(lo, seed, B, K, D):
    """The op sequence of EncoderNewsVAE.encode after the transformer (modules/encoder.py:79-107) with
    injected eps, combined into a beta-TC-VAE style scalar (MI, TC, dimension-wise KL with weights, cf.
    loss_and_optimisation.py:816-824,:925-966) plus the analytic KL, and its gradients w.r.t. mu, logvar."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(B, D, generator=g).requires_grad_(True)
    logvar = (0.5 * torch.randn(B, D, generator=g) - 1.0).requires_grad_(True)
    eps = torch.randn(B, K, D, generator=g)
    std = logvar.mul(0.5).exp()
    latent_z = mu.unsqueeze(1) + eps * std.unsqueeze(1)                       # reparameterize :148-157
    single = latent_z[:, 0, :]
    log_q_z_x = lo.sample_log_likelihood(latent_z, mu=mu, logvar=logvar, reduce_latent_dim=True,
                                         reduce_batch_dim=False)
    log_p_z = lo.sample_log_likelihood(latent_z, reduce_latent_dim=True, reduce_batch_dim=False)
    log_q_z, log_q_z_pm = lo.approximate_log_q_z(mu, logvar, single, method="chen", dataset_size=10000,
                                                 prod_marginals=True)
    kl, hinge = lo.kl_divergence(mu, logvar, hinge_kl_loss_lambda=0.5)
    mi = log_q_z_x.mean() - log_q_z
    tc = log_q_z - log_q_z_pm
    dim_kl = log_q_z_pm - log_p_z.mean()
    loss = 1.0 * mi + 3.0 * tc + 0.5 * dim_kl + 0.25 * kl + 0.1 * hinge
    loss.backward()
    out = {"mu": np32(mu), "logvar": np32(logvar), "eps": np32(eps), "loss": np32(loss), "mi": np32(mi),
           "tc": np32(tc), "dim_kl": np32(dim_kl), "kl": np32(kl), "hinge": np32(hinge),
           "d_mu": np32(mu.grad), "d_logvar": np32(logvar.grad)}
    np.savez_compressed(os.path.join(OUT, "grad_encode_tail.npz"), **out)
    print("wrote grad_encode_tail")


def case_mi_generative(mi, name, seed, K, V, peaked):
    """calc_upper_and_lower_bound_generative_mi (mutual_information.py:251-278) on [K, 1, V] probabilities
    (seq_len = 1, the only shape the function broadcasts for).  The multinomial draw inside the function
    is reproduced by re-seeding the global generator before an identical call."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(K, V, generator=g) * (4.0 if peaked else 1.0)
    if peaked:                                 # each row favours "its own" class: a positive MI
        logits[torch.arange(K), torch.arange(K) % V] += 6.0
    probs = torch.softmax(logits, dim=1)
    probs[0, : V // 4] = 0.0                   # exact zeros: exercises the eps clamp of probs_to_logits
    probs = probs * (0.5 + torch.rand(K, 1, generator=g))      # un-normalised rows are allowed by torch
    batch_probs = probs.unsqueeze(1).contiguous()
    torch.manual_seed(seed + 1000)
    draw = torch.multinomial(batch_probs.reshape(-1, V), num_samples=1).squeeze(1)
    torch.manual_seed(seed + 1000)
    lower, upper = mi.calc_upper_and_lower_bound_generative_mi(batch_probs)
    dists = [torch.distributions.Categorical(probs=probs[i]) for i in range(K)]
    pairs = mi.log_prob_pairs(dists, draw)
    out = {"batch_probs": np32(batch_probs), "sampled_predictions": draw.numpy().astype(np.int64),
           "lower": np.float64(lower), "upper": np.float64(upper), "log_prob_pairs": np32(pairs)}
    np.savez_compressed(os.path.join(OUT, f"mi_generative_{name}.npz"), **out)
    print("wrote mi_generative", name, lower, upper)


def case_pareto(seed):
    """is_pareto_efficient_simple (utils_train.py:416-431).  utils_train does not import here (it pulls transformers
    4.2.2 internals at module scope), but the function is sixteen self-contained lines of NumPy: its FunctionDef node
    is taken out of the reference file with `ast` and run as the reference wrote it.  Gain matrices with ties,
    duplicates, a dominated copy placed BEFORE its dominator, one column, and a single point."""
    import ast
    path = os.path.join(REF, "utils_train.py")
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "is_pareto_efficient_simple")
    ns = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)     # the reference's own function
    fn = ns["is_pareto_efficient_simple"]
    rng = np.random.default_rng(seed)
    out = {}
    shapes = [(1, 3), (12, 2), (40, 3), (25, 5), (300, 2), (64, 7), (50, 1), (200, 4)]
    for k, (n, d) in enumerate(shapes):
        g = np.round(rng.standard_normal((n, d)), 1 if k % 2 == 0 else 3)        # coarse rounding: ties and duplicates
        if n > 4:
            g[n // 2] = g[0]                     # exact duplicate of an earlier point
            g[1] = g[n - 1] - 0.5                # dominated point ahead of its dominator
            g[n // 3, 0] = g[n // 3 + 1, 0]      # tie in one coordinate
        out[f"gains_{k}"] = g.astype(np.float64)
        out[f"efficient_{k}"] = fn(g.copy())
    out["n_cases"] = np.array(len(shapes))
    np.savez_compressed(os.path.join(OUT, "pareto.npz"), **out)
    print("pareto", [int(out[f"efficient_{k}"].sum()) for k in range(len(shapes))])


def main():
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return
    torch.set_num_threads(8)
    lo, kde, enc, mi = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "--only-wide-grad":
        case_grad(lo, "d100", 28, 40, 56, 100)
        case_grad(lo, "d128", 29, 32, 32, 128)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-generative-mi":
        case_mi_generative(mi, "k64_v500", 26, 64, 500, False)
        case_mi_generative(mi, "k128_v600_peaked", 27, 128, 600, True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-pareto":
        case_pareto(32)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-round2":
        case_kde_batch(kde, 30)
        case_gll_generic(mi, 31)
        return
    case_logqz(lo, "generic", 0, 256, 256, 32, "generic")
    case_logqz(lo, "ragged", 1, 100, 77, 32, "generic")
    case_logqz(lo, "d64", 2, 128, 128, 64, "generic")
    case_logqz(lo, "collapsed", 3, 64, 64, 32, "collapsed")
    case_logqz(lo, "near_collapsed", 4, 192, 192, 32, "near_collapsed")
    case_logqz(lo, "peaked", 5, 128, 128, 32, "peaked")
    case_logqz(lo, "far", 6, 96, 96, 32, "far")
    case_logqz(lo, "d20", 7, 50, 30, 20, "generic")
    case_logqz(lo, "d16", 8, 70, 70, 16, "generic")
    case_logqz(lo, "n1", 9, 1, 1, 32, "generic")
    case_mi(lo, mi, "k128", 10, 128, 32)
    case_mi(lo, mi, "k16", 11, 16, 32)
    case_qz_estimate(mi, 12, 100, 256, 32)
    case_iw(lo, enc, "k1", 13, 64, 1, 32)
    case_iw(lo, enc, "k600", 14, 8, 600, 32)
    case_iw(lo, enc, "k37_d64", 15, 9, 37, 64)
    case_kde(lo, kde, 16)
    case_mmd(lo, 17, 256, 32, "n256")
    case_mmd(lo, 18, 100, 64, "n100_d64")
    case_grad(lo, "b32", 19, 32, 32, 32)
    case_grad(lo, "ragged_d20", 20, 45, 70, 20)
    case_grad(lo, "d64", 21, 96, 64, 64)
    case_grad(lo, "n300", 22, 300, 300, 32)
    case_grad(lo, "near_collapsed", 23, 64, 64, 32, "near_collapsed")
    case_grad_elem(lo, 24, 12, 5, 32)
    case_grad_encode_tail(lo, 25, 32, 3, 32)
    case_mi_generative(mi, "k64_v500", 26, 64, 500, False)
    case_mi_generative(mi, "k128_v600_peaked", 27, 128, 600, True)
    case_grad(lo, "d100", 28, 40, 56, 100)
    case_grad(lo, "d128", 29, 32, 32, 128)
    case_kde_batch(kde, 30)
    case_gll_generic(mi, 31)
    case_pareto(32)


if __name__ == "__main__":
    main()
