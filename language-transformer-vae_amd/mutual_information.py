"""Mirror of the estimator functions of NewsVAE/mutual_information.py over the HIP kernels."""
import math

import torch

from . import _native as N


def _spread_pairs(z, mu, spread, spread_kind):
    z, mu, spread = N.as_f32(z, "sample"), N.as_f32(mu, "mu"), N.as_f32(spread, "var")
    M, D = z.shape
    Nn = mu.shape[0]
    dev = z.device
    with torch.cuda.device(dev):
        out = torch.empty(Nn, M, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_log_prob_pairs_f32(N.ptr(z), N.ptr(mu), N.ptr(spread), spread_kind, M, Nn, D,
                                                N.ptr(out), N.stream_ptr(dev)))
    return out


def _gaussian_log_likelihood(sample, mask=None, dim=2, mu=None, var=None):
    """mutual_information.py:26-41.  The element-wise Gaussian term over the broadcast shape of (sample, mu,
    var), weighted with ``mask.unsqueeze(dim)`` and summed over ``dim``; missing mu / var mean standard ones
    (:32-39).  The layout q_z_estimate uses -- sample [S,1,D] against [N,D] parameters, mask [[1.]], dim 2 ->
    [S,N] -- goes through the pair-matrix kernel; every other mask / axis / layout through
    lvae_gauss_ll_masked_f32 (operands passed as broadcast strides, nothing materialised)."""
    if mask is None:
        mask = torch.ones(1, 1, device=sample.device)
    pair_layout = (dim == 2 and sample.dim() == 3 and sample.shape[1] == 1 and mask.numel() == 1
                   and (mu is None or mu.dim() == 2) and (var is None or var.dim() == 2)
                   and (mu is not None or var is not None) and float(mask.reshape(-1)[0]) == 1.0)
    if pair_layout:
        s = sample.reshape(sample.shape[0], sample.shape[-1])
        if mu is None:
            mu_, var_ = torch.zeros_like(var), var
        elif var is None:
            mu_, var_ = mu, torch.ones_like(mu)
        else:
            mu_, var_ = mu, var
        return _spread_pairs(s, mu_, var_, N.SPREAD_VAR).t().contiguous()
    s = N.as_f32(sample, "sample")
    ops = [s] + [N.as_f32(t, n) for t, n in ((mu, "mu"), (var, "var")) if t is not None]
    full = torch.broadcast_shapes(*[t.shape for t in ops])
    nd = len(full)
    if nd < 1 or nd > 6:
        raise ValueError("the broadcast shape of sample / mu / var must have 1..6 axes")
    ax = dim if dim >= 0 else nd + dim
    if not 0 <= ax < nd:
        raise IndexError(f"dim {dim} out of range for the broadcast shape {tuple(full)}")
    m = N.as_f32(mask, "mask").unsqueeze(dim)           # :34-41, mask.unsqueeze(dim)
    full = torch.broadcast_shapes(full, m.shape)
    if len(full) != nd:
        raise ValueError("mask.unsqueeze(dim) must not add axes to the broadcast shape")
    arr = lambda v: (N._i64 * nd)(*[int(x) for x in v])
    strides = lambda t: None if t is None else arr(N.as_f32(t, "operand").expand(full).stride())
    mu_c = None if mu is None else N.as_f32(mu, "mu")
    var_c = None if var is None else N.as_f32(var, "var")
    out_shape = [full[k] for k in range(nd) if k != ax]
    dev = s.device
    with torch.cuda.device(dev):
        out = torch.empty(out_shape, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_gauss_ll_masked_f32(N.ptr(s), N.ptr(mu_c), N.ptr(var_c), N.ptr(m), nd, arr(full),
                                                 strides(s), strides(mu_c), strides(var_c), strides(m), ax,
                                                 N.ptr(out), N.stream_ptr(dev)))
    return out


def q_z_estimate(z, mu, var, device=None):
    """mutual_information.py:44-52 including quirk Q2: the logsumexp runs over dim=1 of [1,S,N], the
    SAMPLE axis, so the result is [1,N]: LSE_i log q(z_i|x_j) - log N for every posterior j.
    (The reference's debug prints, :49-50, are not reproduced.)"""
    z, mu, var = N.as_f32(z, "z"), N.as_f32(mu, "mu"), N.as_f32(var, "var")
    M, D = z.shape
    Nn = mu.shape[0]
    dev = z.device
    if D > 128:
        # wider than the fused kernel's operand registers: the explicit [N, M] matrix (lvae_log_prob_pairs_f32)
        # reduced row by row with the IW logsumexp kernel (LSE_k - log K); moderate sizes only
        if M * Nn > (1 << 28):
            raise N.LvaeError(f"q_z_estimate with D > 128 materialises the [N, M] matrix: N * M <= 2^28 (got {Nn} x {M})")
        P = _spread_pairs(z, mu, var, N.SPREAD_VAR)
        with torch.cuda.device(dev):
            zeros = torch.zeros_like(P)
            out = torch.empty(Nn, dtype=torch.float32, device=dev)
            N.check(N.lib().lvae_iw_lse_f32(N.ptr(P), N.ptr(zeros), N.ptr(zeros), Nn, M, N.ptr(out),
                                            N.stream_ptr(dev)))
        return (out + (math.log(M) - math.log(Nn))).unsqueeze(0)
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_log_qz_workspace_bytes(Nn, M, D, 0, nbytes))    # roles swapped
        ws = N.workspace(nbytes.value, dev)
        out = torch.empty(Nn, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_qz_estimate_f32(N.ptr(z), N.ptr(mu), N.ptr(var), N.SPREAD_VAR, M, Nn, D,
                                             math.log(Nn), N.ptr(out), N.ptr(ws), nbytes.value,
                                             N.stream_ptr(dev)))
    return out.unsqueeze(0)


def calc_mi_estimate(log_q_z_x, latents, mu, logvar, log_p_z, avg_KL, log_q_z_method="aggregate_posterior",
                     mi_method="zhao"):
    """mutual_information.py:56-105, aggregate_posterior branch: log_q_z = LSE(log_q_z_x) - log N over the
    [N] vector (:85), marg_KL = log_q_z - mean(log_p_z) (:94), MI by Zhao (:98) or Hoffman (:101).
    Returns (mi_estimate, marg_KL) as Python floats.  The reference's dead "kernel" branch (:73-83) is
    not provided."""
    if log_q_z_method != "aggregate_posterior":
        raise NotImplementedError
    lq = N.as_f32(log_q_z_x, "log_q_z_x")
    if lq.dim() != 1:
        raise ValueError("log_q_z_x must be the [N] vector of log q(z_i|x_i) (a matrix fails at .item() "
                         "in the reference too)")
    n = lq.shape[0]
    dev = lq.device
    with torch.cuda.device(dev):          # LSE_k(x_k) - log n is the IW reduction of one [1, n] row
        zeros = torch.zeros(1, n, dtype=torch.float32, device=dev)
        lse = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_iw_lse_f32(N.ptr(lq), N.ptr(zeros), N.ptr(zeros), 1, n, N.ptr(lse),
                                        N.stream_ptr(dev)))
    log_q_z = lse[0].double()
    marg_KL = log_q_z - N.reduce_f64(log_p_z, 0, 1.0 / log_p_z.numel())
    if mi_method == 'zhao':
        mi_estimate = (N.reduce_f64(lq, 0, 1.0 / n) - log_q_z).item()
    elif mi_method == 'hoffman':
        mi_estimate = (float(avg_KL) - marg_KL).item()
    else:
        raise ValueError('MI method {} is unknown. Please choose [zhao, hoffman]'.format(mi_method))
    return mi_estimate, marg_KL.item()


def _dists_to_params(dists):
    """Accept (mu, cov_diag) tensors, or the reference's list of torch distributions
    (MultivariateNormal with diagonal covariance / Normal)."""
    if isinstance(dists, (tuple, list)) and len(dists) == 2 and isinstance(dists[0], torch.Tensor):
        return dists
    mus, covs = [], []
    for d in dists:
        mus.append(d.loc)
        if hasattr(d, "covariance_matrix"):
            covs.append(torch.diagonal(d.covariance_matrix))
        else:
            covs.append(d.scale ** 2)
    return torch.stack(mus), torch.stack(covs)


def log_prob_pairs(dists, sample_batch):
    """mutual_information.py:108-132: [K, K] matrix P[j, i] = log p_j(sample_i); rows = distributions."""
    mu, cov = _dists_to_params(dists)
    return _spread_pairs(sample_batch.squeeze(), mu, cov, N.SPREAD_VAR)


def _bounds(samples, dists, path=N.PATH_AUTO):
    mu, cov = _dists_to_params(dists)
    z, mu, cov = N.as_f32(samples.squeeze(), "samples"), N.as_f32(mu, "mu"), N.as_f32(cov, "cov")
    if z.dim() == 1 and mu.dim() == 2 and mu.shape[1] == 1:
        z = z.unsqueeze(1)          # latent_size 1: the reference's squeeze() (mutual_information.py:127) drops the axis
    K, D = z.shape
    dev = z.device
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_pair_bounds_workspace_bytes(K, D, nbytes))
        ws = N.workspace(nbytes.value, dev)
        out = torch.empty(2, dtype=torch.float64, device=dev)
        N.check(N.lib().lvae_pair_bounds_f32(N.ptr(z), N.ptr(mu), N.ptr(cov), K, D, path, N.ptr(out),
                                             N.ptr(ws), nbytes.value, N.stream_ptr(dev)))
    return out


def mi_lower_bound(samples, dists):
    """mutual_information.py:135-173: InfoNCE lower bound, mean_i(P[i,i] - (LSE_j P[j,i] - log K))."""
    return _bounds(samples, dists)[0].item()


def mi_upper_bound(samples, dists):
    """mutual_information.py:176-215: MBU upper bound, the diagonal masked out of the logsumexp (:207).
    Unlike the reference (which builds the mask on the CPU) this runs where the samples are."""
    return _bounds(samples, dists)[1].item()


def calc_upper_and_lower_bound_representational_mi(batch_latent_samples, batch_mu_logvars, gauss_dists=None):
    """mutual_information.py:218-248.  Quirk Q3 is kept: the distributions are MultivariateNormal(mu,
    diag(std)) (:240-243), i.e. the covariance diagonal is std = exp(0.5 * logvar)."""
    if not gauss_dists:
        mu, logvar = torch.chunk(batch_mu_logvars, 2, dim=1)
        std = torch.exp(0.5 * logvar)
        gauss_dists = (mu.contiguous(), std.contiguous())
    lower, upper = _bounds(batch_latent_samples, gauss_dists).tolist()
    return lower, upper


def _categorical_bounds(probs, samples):
    probs = N.as_f32(probs, "batch_probs")
    K, V = probs.shape
    dev = probs.device
    samples = samples.to(device=dev, dtype=torch.int64).contiguous()
    if samples.shape != (K,):
        raise RuntimeError(f"need one sampled prediction per distribution: {tuple(samples.shape)} vs K={K}")
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_categorical_bounds_workspace_bytes(K, nbytes))
        ws = N.workspace(nbytes.value, dev)
        out = torch.empty(2, dtype=torch.float64, device=dev)
        N.check(N.lib().lvae_categorical_bounds_f32(N.ptr(probs), N.ptr(samples), K, V, N.ptr(out), N.ptr(ws),
                                                    nbytes.value, N.stream_ptr(dev)))
    return out


def calc_upper_and_lower_bound_generative_mi(batch_probs, cat_dists=None, sampled_predictions=None):
    """mutual_information.py:251-278: InfoNCE / MBU bounds with the decoder's Categorical distributions as
    critics.  batch_probs [batch, seq_len, vocab]; one class is drawn per row (torch.multinomial on the
    caller's device, :269) unless ``sampled_predictions`` injects the draw (tests; not in the reference).
    ``cat_dists``: the reference's optional list of torch Categorical objects (their ``probs`` are used).

    As written the reference only broadcasts for seq_len == 1: it builds K = batch distributions from the
    first K rows of the flattened [batch * seq_len, vocab] matrix (:271-273) but draws batch * seq_len
    samples, and subtracts a [K * seq_len] logsumexp from the [K] diagonal (:165-171).  The same condition
    is enforced here with the error the reference would raise (RuntimeError)."""
    K = batch_probs.shape[0]
    flat = batch_probs.reshape(-1, batch_probs.shape[-1])
    if flat.shape[0] != K:
        raise RuntimeError(f"The size of tensor a ({K}) must match the size of tensor b ({flat.shape[0]}) at "
                           "non-singleton dimension 0 (the reference supports seq_len == 1 only)")
    if not flat.is_cuda:
        raise N.LvaeError("batch_probs must be a CUDA (HIP) tensor: there is no CPU fallback")
    if sampled_predictions is None:
        sampled_predictions = torch.multinomial(flat, num_samples=1).squeeze(1)
    if cat_dists:
        flat = torch.stack([d.probs for d in cat_dists]).to(flat.device)
    lower, upper = _categorical_bounds(flat, sampled_predictions).tolist()
    return lower, upper
