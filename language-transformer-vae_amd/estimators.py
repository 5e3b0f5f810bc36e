"""Host-side mirror of the reference's latent estimators (NewsVAE/loss_and_optimisation.py).

Same names, positional order, defaults and return shapes as the reference functions cited on each
definition; the bodies call the HIP kernels through the C ABI (``_native``).  The estimators the reference
differentiates in its training objectives (approximate_log_q_z, sample_log_likelihood, kl_divergence,
reparameterisation; loss_and_optimisation.py:925-1018) carry HIP backward passes behind torch.autograd.Function;
the others are forward-only and say so when handed inputs that require grad while grad mode is on.
"""
import math

import torch

from . import _native as N


def _pair_workspace(M, Nn, D, with_pm, device):
    nbytes = N._i64(0)
    N.check(N.lib().lvae_log_qz_workspace_bytes(M, Nn, D, int(with_pm), nbytes))
    return N.workspace(nbytes.value, device), nbytes.value


def log_qz_raw(mu, spread, latent_z, log_norm, prod_marginals=False, spread_kind=N.SPREAD_LOGVAR,
               exclude_diag=False, diag_offset=0, path=N.PATH_AUTO, want_joint=True):
    """log q(z_i) (and optionally the product-of-marginals term) for every row of ``latent_z`` against
    all posteriors (mu, spread).  Returns ([M] or None, [M] or None)."""
    mu, spread, z = N.as_f32(mu, "mu"), N.as_f32(spread, "logvar"), N.as_f32(latent_z, "latent_z")
    if mu.dim() != 2 or z.dim() != 2 or mu.shape != spread.shape or mu.shape[1] != z.shape[1]:
        raise ValueError(f"shape mismatch: mu {tuple(mu.shape)}, logvar {tuple(spread.shape)}, "
                         f"latent_z {tuple(z.shape)}")
    Nn, D = mu.shape
    M = z.shape[0]
    dev = z.device
    with torch.cuda.device(dev):
        ws, nbytes = _pair_workspace(M, Nn, D, prod_marginals, dev)
        out = torch.empty(M, dtype=torch.float32, device=dev) if want_joint else None
        out_pm = torch.empty(M, dtype=torch.float32, device=dev) if prod_marginals else None
        N.check(N.lib().lvae_log_qz_f32(N.ptr(z), N.ptr(mu), N.ptr(spread), spread_kind, M, Nn, D,
                                        float(log_norm), int(exclude_diag), int(diag_offset), path,
                                        N.ptr(out), N.ptr(out_pm), N.ptr(ws), nbytes, N.stream_ptr(dev)))
    return out, out_pm


def log_qz_partial(mu, spread, latent_z, spread_kind=N.SPREAD_LOGVAR, path=N.PATH_AUTO):
    """Partial logsumexp of ``latent_z`` [M, D] over the given SUBSET of posteriors [n, D]
    (lvae_log_qz_partial_f32): (row_max, row_sum), natural-log domain, log q = row_max + log(row_sum) - log N
    once every subset has been merged with ``lse_combine``."""
    mu, spread, z = N.as_f32(mu, "mu"), N.as_f32(spread, "logvar"), N.as_f32(latent_z, "latent_z")
    if mu.dim() != 2 or z.dim() != 2 or mu.shape != spread.shape or mu.shape[1] != z.shape[1]:
        raise ValueError(f"shape mismatch: mu {tuple(mu.shape)}, logvar {tuple(spread.shape)}, "
                         f"latent_z {tuple(z.shape)}")
    Nn, D = mu.shape
    M = z.shape[0]
    dev = z.device
    with torch.cuda.device(dev):
        ws, nbytes = _pair_workspace(M, Nn, D, False, dev)
        row_max = torch.empty(M, dtype=torch.float32, device=dev)
        row_sum = torch.empty(M, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_log_qz_partial_f32(N.ptr(z), N.ptr(mu), N.ptr(spread), spread_kind, M, Nn, D, path,
                                                N.ptr(row_max), N.ptr(row_sum), N.ptr(ws), nbytes,
                                                N.stream_ptr(dev)))
    return row_max, row_sum


def lse_combine(a, b=None, log_norm=None):
    """Merge two partial logsumexps (pairs from ``log_qz_partial``).  With ``log_norm`` the finished
    log q(z) = max + log(sum) - log_norm is returned, otherwise the merged (max, sum) pair."""
    ma, sa = (N.as_f32(t, "partial") for t in a)
    mb, sb = (N.as_f32(t, "partial") for t in b) if b is not None else (None, None)
    n = ma.numel()
    dev = ma.device
    with torch.cuda.device(dev):
        if log_norm is not None:
            out = torch.empty(n, dtype=torch.float32, device=dev)
            N.check(N.lib().lvae_lse_combine_f32(N.ptr(ma), N.ptr(sa), N.ptr(mb), N.ptr(sb), n, float(log_norm),
                                                 N.ptr(out), None, None, N.stream_ptr(dev)))
            return out
        om, os_ = torch.empty_like(ma), torch.empty_like(sa)
        N.check(N.lib().lvae_lse_combine_f32(N.ptr(ma), N.ptr(sa), N.ptr(mb), N.ptr(sb), n, 0.0, None,
                                             N.ptr(om), N.ptr(os_), N.stream_ptr(dev)))
    return om, os_


class _LogQz(torch.autograd.Function):
    """log q(z) / prod-marginals with a HIP backward (lvae_log_qz_bwd_f32): the softmax weights are
    recomputed from the saved forward result, like the forward nothing of size M x N is stored."""

    @staticmethod
    def forward(ctx, mu, logvar, z, log_norm, prod_marginals, path):
        mu_d, lv_d, z_d = mu.detach(), logvar.detach(), z.detach()
        lq, lqpm = log_qz_raw(mu_d, lv_d, z_d, log_norm, prod_marginals, path=path)
        ctx.save_for_backward(N.as_f32(mu_d, "mu"), N.as_f32(lv_d, "logvar"), N.as_f32(z_d, "latent_z"), lq)
        ctx.log_norm, ctx.prod_marginals = float(log_norm), bool(prod_marginals)
        if lqpm is None:
            lqpm = lq.new_zeros(())
            ctx.mark_non_differentiable(lqpm)
        return lq, lqpm

    @staticmethod
    def backward(ctx, g_lq, g_pm):
        mu, logvar, z, lq = ctx.saved_tensors
        Nn, D = mu.shape
        M = z.shape[0]
        dev = z.device
        need_mu, need_lv, need_z = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g_lq = None if g_lq is None else N.as_f32(g_lq.detach(), "grad").expand(M).contiguous()
        g_pm = (None if (g_pm is None or not ctx.prod_marginals)
                else N.as_f32(g_pm.detach(), "grad").expand(M).contiguous())
        if g_lq is None and g_pm is None:
            return None, None, None, None, None, None
        with torch.cuda.device(dev):
            nbytes = N._i64(0)
            N.check(N.lib().lvae_log_qz_bwd_workspace_bytes(M, Nn, D, int(g_pm is not None), nbytes))
            ws = N.workspace(nbytes.value, dev)
            d_z = torch.empty_like(z) if need_z else None
            d_mu = torch.empty_like(mu) if need_mu else None
            d_lv = torch.empty_like(logvar) if need_lv else None
            N.check(N.lib().lvae_log_qz_bwd_f32(N.ptr(z), N.ptr(mu), N.ptr(logvar), M, Nn, D, ctx.log_norm,
                                                N.ptr(lq), N.ptr(g_lq), N.ptr(g_pm), N.ptr(d_z), N.ptr(d_mu),
                                                N.ptr(d_lv), N.ptr(ws), nbytes.value, N.stream_ptr(dev)))
        return d_mu, d_lv, d_z, None, None, None


def approximate_log_q_z(mu, logvar, latent_z, method="chen", dataset_size=42068, prod_marginals=False,
                        reduce_mean=True, path=N.PATH_AUTO):
    """loss_and_optimisation.py:139-221.  mu, logvar [x_batch, D]; latent_z [z_batch, D].
    Returns (log_q_z, log_q_z_prod_marg): [z_batch] vectors, or 0-dim means with ``reduce_mean``.
    Quirk Q1 is kept: ``prod_marginals=False, reduce_mean=True`` raises AttributeError like the
    reference (``None.mean()``, :219)."""
    if method == "chen":
        assert dataset_size is not None, "if using method 'chen', you need to provide a dataset size"
    x_batch = mu.shape[0]
    log_norm = math.log(x_batch * dataset_size) if method == "chen" else math.log(x_batch)   # :207,:215
    if _needs_grad(mu, logvar, latent_z):
        # training-time use (the objectives of :925-1018 differentiate this estimator): HIP forward and
        # HIP backward behind torch.autograd (D > 64: training batches only, M * N <= 2^28)
        log_q_z, log_q_z_prod_marg = _LogQz.apply(mu, logvar, latent_z, log_norm, prod_marginals, path)
        if not prod_marginals:
            log_q_z_prod_marg = None
        if reduce_mean:
            if log_q_z_prod_marg is None:                # Q1
                raise AttributeError("'NoneType' object has no attribute 'mean'")
            return log_q_z.mean(), log_q_z_prod_marg.mean()
        return log_q_z, log_q_z_prod_marg
    log_q_z, log_q_z_prod_marg = log_qz_raw(mu, logvar, latent_z, log_norm, prod_marginals, path=path)
    if reduce_mean:
        if log_q_z_prod_marg is None:                    # Q1, :219 -- ``None.mean()``
            raise AttributeError("'NoneType' object has no attribute 'mean'")
        log_q_z = N.mean_f32(log_q_z)
        log_q_z_prod_marg = N.mean_f32(log_q_z_prod_marg)
    return log_q_z, log_q_z_prod_marg


def _sample_ll_args(latent_z, mu, logvar):
    """Validated float32 views and the (B, K, D) geometry of one sample_log_likelihood call."""
    z = N.as_f32(latent_z, "latent_z")
    post = mu is not None and logvar is not None
    D = z.shape[-1]
    if post:
        mu, logvar = N.as_f32(mu, "mu"), N.as_f32(logvar, "logvar")
        if z.dim() == 3:
            mu2 = mu.reshape(mu.shape[0], D) if mu.dim() == 3 and mu.shape[1] == 1 else mu
            lv2 = logvar.reshape(logvar.shape[0], D) if logvar.dim() == 3 and logvar.shape[1] == 1 else logvar
            if mu2.dim() != 2 or mu2.shape != (z.shape[0], D) or lv2.shape != mu2.shape:
                raise ValueError(f"unsupported broadcast: latent_z {tuple(z.shape)}, mu {tuple(mu.shape)}")
            B, K = z.shape[0], z.shape[1]
        elif z.dim() == 2:
            if mu.shape != z.shape or logvar.shape != z.shape:
                raise ValueError(f"unsupported broadcast: latent_z {tuple(z.shape)}, mu {tuple(mu.shape)}")
            mu2, lv2, B, K = mu, logvar, z.shape[0], 1
        else:
            raise ValueError("latent_z must be [B,K,D] or [M,D] for the posterior branch")
        mu2, lv2 = mu2.contiguous(), lv2.contiguous()
    else:
        mu2 = lv2 = None
        B, K = z.numel() // D, 1
    return z, mu2, lv2, B, K, D


def _sample_ll_forward(z, mu2, lv2, B, K, D, reduce_latent_dim):
    dev = z.device
    post = mu2 is not None
    with torch.cuda.device(dev):
        if reduce_latent_dim:
            out = torch.empty(z.shape[:-1], dtype=torch.float32, device=dev)
            N.check(N.lib().lvae_sample_ll_f32(N.ptr(z), N.ptr(mu2), N.ptr(lv2), B, K, D,
                                               N.ptr(out) if post else None,
                                               None if post else N.ptr(out), N.stream_ptr(dev)))
        else:
            out = torch.empty_like(z)
            N.check(N.lib().lvae_sample_ll_elem_f32(N.ptr(z), N.ptr(mu2), N.ptr(lv2), B, K, D, N.ptr(out),
                                                    N.stream_ptr(dev)))
    return out


class _SampleLL(torch.autograd.Function):
    """sample_log_likelihood with a HIP backward (lvae_sample_ll_bwd_f32)."""

    @staticmethod
    def forward(ctx, latent_z, mu, logvar, reduce_latent_dim):
        det = lambda t: None if t is None else t.detach()
        z, mu2, lv2, B, K, D = _sample_ll_args(det(latent_z), det(mu), det(logvar))
        ctx.geom = (B, K, D, bool(reduce_latent_dim), latent_z.shape,
                    None if mu is None else mu.shape, None if logvar is None else logvar.shape)
        ctx.save_for_backward(z, mu2, lv2)
        return _sample_ll_forward(z, mu2, lv2, B, K, D, reduce_latent_dim)

    @staticmethod
    def backward(ctx, g):
        z, mu2, lv2 = ctx.saved_tensors
        B, K, D, red, zshape, mushape, lvshape = ctx.geom
        dev = z.device
        g = N.as_f32(g.detach(), "grad").expand(z.shape[:-1] if red else z.shape).contiguous()
        need_z, need_mu, need_lv = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        post = mu2 is not None
        with torch.cuda.device(dev):
            d_z = torch.empty_like(z) if need_z else None
            d_mu = torch.empty_like(mu2) if (post and need_mu) else None
            d_lv = torch.empty_like(lv2) if (post and need_lv) else None
            if d_z is not None or d_mu is not None or d_lv is not None:
                N.check(N.lib().lvae_sample_ll_bwd_f32(N.ptr(z), N.ptr(mu2), N.ptr(lv2), B, K, D, N.ptr(g),
                                                       0 if red else 1, N.ptr(d_z), N.ptr(d_mu), N.ptr(d_lv),
                                                       N.stream_ptr(dev)))
        return (None if d_z is None else d_z.reshape(zshape), None if d_mu is None else d_mu.reshape(mushape),
                None if d_lv is None else d_lv.reshape(lvshape), None)


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def sample_log_likelihood(latent_z, mu=None, logvar=None, reduce_latent_dim=True, reduce_batch_dim=True):
    """loss_and_optimisation.py:79-110.  Log-density of samples under the posterior N(mu, exp(logvar))
    (both given) or the standard-normal prior (neither given).
    Supported layouts: latent_z [B,K,D] with mu/logvar [B,D] or [B,1,D] (:86-90), latent_z [M,D] with
    mu/logvar [M,D], and any [..., D] for the prior.  Differentiable (HIP backward) when an input
    requires grad."""
    if mu is None or logvar is None:
        mu = logvar = None
    if _needs_grad(latent_z, mu, logvar):
        out = _SampleLL.apply(latent_z, mu, logvar, reduce_latent_dim)
        return out.mean(dim=0) if reduce_batch_dim else out          # :107-108
    z, mu2, lv2, B, K, D = _sample_ll_args(latent_z, mu, logvar)
    out = _sample_ll_forward(z, mu2, lv2, B, K, D, reduce_latent_dim)
    if reduce_batch_dim:                                  # :107-108 mean over dim 0
        return N.mean_f32(out) if out.dim() <= 1 else N.colmean_f32(out)
    return out


def _kl_forward(mu, logvar, lam, sum_latent):
    n, D = mu.shape
    dev = mu.device
    with torch.cuda.device(dev):
        if sum_latent:
            kl = torch.empty(n, dtype=torch.float32, device=dev)
            hk = torch.empty(n, dtype=torch.float32, device=dev)
            N.check(N.lib().lvae_kl_diag_f32(N.ptr(mu), N.ptr(logvar), n, D, float(lam),
                                             None, None, N.ptr(kl), N.ptr(hk), N.stream_ptr(dev)))
        else:
            kl, hk = torch.empty_like(mu), torch.empty_like(mu)
            N.check(N.lib().lvae_kl_diag_f32(N.ptr(mu), N.ptr(logvar), n, D, float(lam),
                                             N.ptr(kl), N.ptr(hk), None, None, N.stream_ptr(dev)))
    return kl, hk


class _KL(torch.autograd.Function):
    """kl_divergence ([N] rows or [N,D] elements) with a HIP backward (lvae_kl_diag_bwd_f32)."""

    @staticmethod
    def forward(ctx, mu, logvar, lam, sum_latent):
        mu_d, lv_d = N.as_f32(mu.detach(), "mu"), N.as_f32(logvar.detach(), "logvar")
        ctx.save_for_backward(mu_d, lv_d)
        ctx.lam, ctx.sum_latent = float(lam), bool(sum_latent)
        return _kl_forward(mu_d, lv_d, lam, sum_latent)

    @staticmethod
    def backward(ctx, g_kl, g_hk):
        mu, logvar = ctx.saved_tensors
        n, D = mu.shape
        dev = mu.device
        shape = (n,) if ctx.sum_latent else (n, D)
        fix = lambda g: None if g is None else N.as_f32(g.detach(), "grad").expand(shape).contiguous()
        g_kl, g_hk = fix(g_kl), fix(g_hk)
        if g_kl is None and g_hk is None:
            return None, None, None, None
        with torch.cuda.device(dev):
            d_mu = torch.empty_like(mu) if ctx.needs_input_grad[0] else None
            d_lv = torch.empty_like(logvar) if ctx.needs_input_grad[1] else None
            if d_mu is not None or d_lv is not None:
                N.check(N.lib().lvae_kl_diag_bwd_f32(N.ptr(mu), N.ptr(logvar), n, D, ctx.lam, N.ptr(g_kl),
                                                     N.ptr(g_hk), int(ctx.sum_latent), N.ptr(d_mu), N.ptr(d_lv),
                                                     N.stream_ptr(dev)))
        return d_mu, d_lv, None, None


def kl_divergence(mu, logvar, hinge_kl_loss_lambda=0.5, average_batch=True, sum_latent=True):
    """loss_and_optimisation.py:14-43.  Returns (kl_loss, hinge_kl_loss) with the reference's shapes:
    0-dim (sum_latent, average_batch), [N] (sum_latent only), [D] (average_batch only) or [N,D].
    Differentiable (HIP backward) when mu or logvar requires grad."""
    if _needs_grad(mu, logvar):
        if mu.dim() != 2 or mu.shape != logvar.shape:
            raise ValueError("mu and logvar must be [batch, latent] of one shape")
        kl, hk = _KL.apply(mu, logvar, hinge_kl_loss_lambda, sum_latent)
        return (kl.mean(dim=0), hk.mean(dim=0)) if average_batch else (kl, hk)
    mu, logvar = N.as_f32(mu, "mu"), N.as_f32(logvar, "logvar")
    if mu.dim() != 2 or mu.shape != logvar.shape:
        raise ValueError("mu and logvar must be [batch, latent] of one shape")
    kl, hk = _kl_forward(mu, logvar, hinge_kl_loss_lambda, sum_latent)
    if average_batch:
        if sum_latent:
            return N.mean_f32(kl), N.mean_f32(hk)
        return N.colmean_f32(kl), N.colmean_f32(hk)
    return kl, hk


def _kernel_matrix(x, y, alphas, dist_eps):
    x, y = N.as_f32(x, "x"), N.as_f32(y, "y")
    if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
        raise ValueError("x and y must be [n, D] and [m, D]")
    dev = x.device
    arr, na = N.host_floats(alphas)
    with torch.cuda.device(dev):
        out = torch.empty(x.shape[0], y.shape[0], dtype=torch.float32, device=dev)
        ws = N.workspace(N.KMAT_WS_BYTES, dev)
        N.check(N.lib().lvae_kernel_matrix_f32(N.ptr(x), N.ptr(y), x.shape[0], y.shape[0], x.shape[1], arr, na,
                                               float(dist_eps), N.ptr(out), N.ptr(ws), N.KMAT_WS_BYTES,
                                               N.stream_ptr(dev)))
    return out


def gaussian_kernel(x, y):
    """loss_and_optimisation.py:63-76: exp(-mean_d((x_i - y_j)^2) / D) = exp(-|x_i - y_j|^2 / D^2), the
    full [n, m] matrix.  Like the reference's ``torch.tensor(...)`` wrap (:76, quirk Q5) the result carries
    no gradient."""
    dim = x.shape[1]
    return _kernel_matrix(x.detach(), y.detach(), [1.0 / float(dim * dim)], 0.0)


def mmd_rbf(x, y, alphas, dist_eps, unbiased, path=N.PATH_AUTO):
    """Gaussian-kernel MMD between samples x [n1,D] and y [n2,D] without materialising the kernel
    matrices (fused MFMA distance + kernel-sum pass).  Returns a float64 device tensor
    [statistic, sum Kxx, sum Kyy, sum Kxy].  ``path``: contraction of the distance (include/lvae.h; AUTO: fp32 MFMA on
    small problems, the gated split-bf16 form from 2^27 pairs on)."""
    x, y = N.as_f32(x, "x"), N.as_f32(y, "y")
    if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
        raise ValueError("x and y must be [n1, D] and [n2, D]")
    n1, n2, D = x.shape[0], y.shape[0], x.shape[1]
    dev = x.device
    arr, na = N.host_floats(alphas)
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_mmd_workspace_bytes(n1, n2, D, nbytes))
        ws = N.workspace(nbytes.value, dev)
        out = torch.empty(4, dtype=torch.float64, device=dev)
        N.check(N.lib().lvae_mmd_rbf_f32(N.ptr(x), N.ptr(y), n1, n2, D, arr, na, float(dist_eps),
                                         int(bool(unbiased)), int(path), N.ptr(out), N.ptr(ws), nbytes.value,
                                         N.stream_ptr(dev)))
    return out


def kernel_sum(a, b, alphas, dist_eps, exclude_diag=False, diag_offset=0, colsum=False, symmetric=False,
               path=N.PATH_AUTO):
    """sum_{i,j} k(a_i, b_j) with k = sum_alpha exp(-alpha (dist_eps + |a_i - b_j|^2)), optionally
    without the pairs j == i + diag_offset: one block of the MMD's kernel-matrix sums (the multi-GPU
    building block; the matrix is never materialised).  Returns a float64 device scalar, and with
    ``colsum`` also the [na] float64 per-row-of-a sums.  ``symmetric``: ``a`` is rows diag_offset.. of ``b``
    (a multiple of 32) and only the upper triangle of tiles is evaluated; the totals of all owners still add
    up to the full sum."""
    a, b = N.as_f32(a, "a"), N.as_f32(b, "b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("a and b must be [na, D] and [nb, D]")
    na, nb, D = a.shape[0], b.shape[0], a.shape[1]
    dev = a.device
    arr, nalpha = N.host_floats(alphas)
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_kernel_sum_workspace_bytes(na, nb, D, nbytes))
        ws = N.workspace(nbytes.value, dev)
        total = torch.empty(1, dtype=torch.float64, device=dev)
        cs = torch.empty(na, dtype=torch.float64, device=dev) if colsum else None
        N.check(N.lib().lvae_kernel_sum_f32(N.ptr(a), N.ptr(b), na, nb, D, arr, nalpha, float(dist_eps),
                                            int(bool(exclude_diag)), int(diag_offset), int(bool(symmetric)),
                                            int(path), N.ptr(total),
                                            N.ptr(cs), N.ptr(ws), nbytes.value, N.stream_ptr(dev)))
    return (total[0], cs) if colsum else total[0]


def mmd_permutation_sums(cols, col_offset, pooled, labels, alphas, dist_eps, symmetric=None, path=N.PATH_AUTO):
    """Label-permutation sums of the pooled kernel matrix for the columns this caller owns
    (lvae_mmd_perm_f32): returns (q [P], lr [P], T) as float64 device tensors.  ``labels`` is a [P, n]
    uint8 device tensor (non-zero: element belongs to sample 1 under that labeling); any P (the C entry
    point takes 128 labelings per pass).  ``symmetric`` evaluates only the tiles on and above the
    diagonal (col_offset must be a multiple of 32); the shares of all owners still add up to the same
    totals PROVIDED every owner uses the same mode.  Default: symmetric iff the call owns all columns."""
    cols, pooled = N.as_f32(cols, "cols"), N.as_f32(pooled, "pooled")
    if symmetric is None:
        symmetric = int(col_offset) == 0 and cols.shape[0] == pooled.shape[0]
    if labels.dtype != torch.uint8 or labels.dim() != 2 or labels.shape[1] != pooled.shape[0]:
        raise ValueError("labels must be a [P, n] uint8 tensor")
    if not labels.is_cuda:
        raise N.LvaeError("labels must live on the GPU")
    labels = labels.contiguous()
    nc, n, D, P = cols.shape[0], pooled.shape[0], pooled.shape[1], labels.shape[0]
    dev = pooled.device
    arr, nalpha = N.host_floats(alphas)
    q = torch.empty(P, dtype=torch.float64, device=dev)
    lr = torch.empty(P, dtype=torch.float64, device=dev)
    tot = None
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_mmd_perm_workspace_bytes(nc, n, D, nbytes))
        ws = N.workspace(nbytes.value, dev)
        for p0 in range(0, P, 128):
            lab = labels[p0:p0 + 128]
            pp = lab.shape[0]
            out = torch.empty(2 * pp + 1, dtype=torch.float64, device=dev)
            N.check(N.lib().lvae_mmd_perm_f32(N.ptr(cols), nc, int(col_offset), N.ptr(pooled), n, D, N.ptr(lab),
                                              pp, arr, nalpha, float(dist_eps), int(bool(symmetric)), int(path),
                                              N.ptr(out), N.ptr(ws), nbytes.value, N.stream_ptr(dev)))
            q[p0:p0 + pp] = out[:pp]
            lr[p0:p0 + pp] = out[pp:2 * pp]
            tot = out[2 * pp]
    return q, lr, tot


def permutation_labels(n1, n2, n_permutations, device, generator=None):
    """[1 + n_permutations, n1 + n2] uint8: row 0 is the observed labeling (first n1 elements in sample
    1), the others are uniformly random relabelings with exactly n1 ones.  All relabelings of a chunk come from ONE
    keyed sort of a [rows, n] matrix of float64 uniforms (the n1 smallest keys of a row carry the ones) instead of one
    sort per labeling: at n = 524,288, P = 100 a single segmented sort, 1,900 launches fewer."""
    n = n1 + n2
    labels = torch.zeros(1 + n_permutations, n, dtype=torch.uint8, device=device)
    labels[0, :n1] = 1
    rows = max(1, min(n_permutations, (1 << 26) // max(n, 1)))      # <= 2^26 keys (0.5 GB of float64) per sort
    for p0 in range(0, n_permutations, rows):
        r = min(rows, n_permutations - p0)
        keys = torch.rand(r, n, dtype=torch.float64, device=device, generator=generator)
        first = keys.argsort(dim=1)[:, :n1]
        labels[1 + p0:1 + p0 + r].scatter_(1, first, 1)
    return labels


def permutation_statistics(q, lr, tot, n1, n2):
    """Statistic of every labeling from its (q, lr) and the total T (float64 tensors)."""
    k11 = q
    k12 = lr - q
    k22 = tot - 2.0 * lr + q
    return k11 / (n1 * (n1 - 1.0)) + k22 / (n2 * (n2 - 1.0)) - 2.0 * k12 / (float(n1) * float(n2))


def mmd_permutation_test(sample_1, sample_2, alphas, n_permutations=1000, dist_eps=1e-5, labels=None,
                         generator=None, path=N.PATH_AUTO):
    """Two-sample permutation test on the unbiased Gaussian-kernel MMD (torch_two_sample
    MMDStatistic.pval restated without the n x n matrix; PARITY UNPINNED).  Returns a dict: ``statistic``
    (observed), ``statistics`` [n_permutations] (float64, one per relabeling) and ``pval`` = fraction of
    relabelings whose statistic is >= the observed one.  ``labels`` ([P, n] uint8, row 0 = observed) may
    be injected for reproducibility."""
    s1, s2 = N.as_f32(sample_1, "sample_1"), N.as_f32(sample_2, "sample_2")
    n1, n2 = s1.shape[0], s2.shape[0]
    pooled = torch.cat((s1, s2), 0).contiguous()
    if labels is None:
        labels = permutation_labels(n1, n2, n_permutations, pooled.device, generator)
    q, lr, tot = mmd_permutation_sums(pooled, 0, pooled, labels, alphas, dist_eps, path=path)
    stats = permutation_statistics(q, lr, tot, n1, n2)
    return {"statistic": stats[0], "statistics": stats[1:], "pval": (stats[1:] >= stats[0]).double().mean()}


def maximum_mean_discrepancy(latent_z, prior_sample=None):
    """loss_and_optimisation.py:46-60: biased RBF MMD between a standard-normal draw x and latent_z:
    mean(Kxx) + mean(Kyy) - 2 mean(Kxy).  The reference draws ``x = randn_like(latent_z)`` inside (:53);
    pass ``prior_sample`` to inject it (parity tests), otherwise it is drawn here the same way."""
    latent_z = latent_z.detach()      # quirk Q5: the reference's kernel detaches (torch.tensor(...), :76) -> no gradient
    x = torch.randn_like(latent_z) if prior_sample is None else prior_sample.detach()
    D = latent_z.shape[1]
    return mmd_rbf(x, latent_z, [1.0 / float(D * D)], 0.0, unbiased=False)[0].float()


def get_tts_mmmd(latent_z, prior_sample=None):
    """LossTermManager.get_tts_mmmd (loss_and_optimisation.py:595-623): torch-two-sample's unbiased MMD
    between latent_z (squeezed, :596) and a standard-normal draw of the same shape (:597), bandwidths
    alphas = [0.1*i for i in range(15)] (:598).  PARITY UNPINNED (third-party source absent)."""
    from .two_sample import MMDStatistic
    latent_z = latent_z.squeeze(1)
    if prior_sample is None:
        prior_sample = torch.randn(latent_z.shape).to(latent_z.device)
    alphas = [0.1 * i for i in range(15)]
    n_1, n_2 = len(latent_z), len(prior_sample)
    return MMDStatistic(n_1, n_2)(latent_z, prior_sample, alphas, ret_matrix=False)


def approximate_marginal_KL(mu, logvar, latent_z, method="chen", dataset_size=None, strict=False):
    """loss_and_optimisation.py:113-121.  As written the reference always raises here (quirk Q1: it asks
    approximate_log_q_z for ``prod_marginals=False`` with the default ``reduce_mean=True``, :114-115,:219).
    ``strict=True`` reproduces that; the default returns the value the code is after:
    mean_i(log q(z_i) - log p(z_i))."""
    if strict:
        approximate_log_q_z(mu, logvar, latent_z, method=method, dataset_size=dataset_size,
                            prod_marginals=False)
    log_q_z, _ = approximate_log_q_z(mu, logvar, latent_z, method=method, dataset_size=dataset_size,
                                     prod_marginals=False, reduce_mean=False)
    log_p_z = sample_log_likelihood(latent_z, reduce_latent_dim=True, reduce_batch_dim=False)
    if log_q_z.requires_grad or log_p_z.requires_grad:       # autograd outputs: torch's mean carries the graph
        return log_q_z.mean() - log_p_z.mean()
    return N.mean_f32(log_q_z) - N.mean_f32(log_p_z)


def approximate_total_correlation(mu, logvar, latent_z, method="chen", dataset_size=None):
    """loss_and_optimisation.py:124-136: E[log q(z)] - E[log prod_d q(z_d)]."""
    log_q_z, log_q_z_prod_marginals = approximate_log_q_z(mu, logvar, latent_z, method=method,
                                                          dataset_size=dataset_size, prod_marginals=True)
    return log_q_z - log_q_z_prod_marginals


def get_1d_kde_dim_marg_kl_estimate(latents, log_p_z, bw="scott", device=None):
    """LossTermManager.get_1d_kde_dim_marg_kl_estimate (loss_and_optimisation.py:625-637): every latent
    coordinate scored under a 1-D Gaussian KDE fitted on all B*D coordinates."""
    from .kde import GaussianKDE
    latents_1D = latents.reshape(-1, 1)
    kde = GaussianKDE(X=latents_1D, bw=bw, device=device)
    log_q_z_kde = kde.score_samples(Y=latents_1D)                      # [B*D]
    # .reshape(latents.shape).sum(dim=1).mean(dim=0) (:630-631) == sum of all scores / B
    log_q_z_kde = N.reduce_f64(log_q_z_kde, 0, 1.0 / latents.shape[0]).float()
    return log_q_z_kde - log_p_z


def iw_log_p_x(log_p_x_z, log_p_z, log_q_z_x):
    """LossTermManager.iw_log_p_x (loss_and_optimisation.py:783-794; twin utils_evaluation.py:176-180):
    logsumexp_k(log_p_x_z + log_p_z - log_q_z_x) - log K.  All inputs [batch, n_samples] -> [batch]."""
    a, b, c = (N.as_f32(t, n) for t, n in ((log_p_x_z, "log_p_x_z"), (log_p_z, "log_p_z"),
                                           (log_q_z_x, "log_q_z_x")))
    if a.dim() != 2 or a.shape != b.shape or a.shape != c.shape:
        raise ValueError("log_p_x_z, log_p_z, log_q_z_x must share one [batch, n_samples] shape")
    B, K = a.shape
    dev = a.device
    with torch.cuda.device(dev):
        out = torch.empty(B, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_iw_lse_f32(N.ptr(a), N.ptr(b), N.ptr(c), B, K, N.ptr(out), N.stream_ptr(dev)))
    return out


def reparam_iw(mu, logvar, eps, log_p_x_z=None, return_z=True, return_densities=True):
    """Fused reparameterize + log q(z|x) + log p(z) + IW log-likelihood (modules/encoder.py:148-157,
    :93-98; loss_and_optimisation.py:783-794) on injected noise ``eps`` [B,K,D].
    Returns dict(latent_z, log_q_z_x, log_p_z, iw_ll).  ``return_densities=False``: only the reduction
    (iw_ll [B]; the two [B,K] density matrices are neither allocated nor written -- 131 MB of writes at
    16,384 x 1,000 that cost the stream a sixth of its time)."""
    mu, logvar, eps = N.as_f32(mu, "mu"), N.as_f32(logvar, "logvar"), N.as_f32(eps, "eps")
    if eps.dim() != 3 or mu.shape != (eps.shape[0], eps.shape[2]) or logvar.shape != mu.shape:
        raise ValueError("need eps [B,K,D] and mu/logvar [B,D]")
    B, K, D = eps.shape
    dev = eps.device
    lpxz = None if log_p_x_z is None else N.as_f32(log_p_x_z, "log_p_x_z")
    if lpxz is not None and lpxz.shape != (B, K):
        raise ValueError("log_p_x_z must be [B,K]")
    with torch.cuda.device(dev):
        z = torch.empty_like(eps) if return_z else None
        lq = torch.empty(B, K, dtype=torch.float32, device=dev) if return_densities else None
        lp = torch.empty(B, K, dtype=torch.float32, device=dev) if return_densities else None
        iw = torch.empty(B, dtype=torch.float32, device=dev)
        N.check(N.lib().lvae_reparam_iw_f32(N.ptr(mu), N.ptr(logvar), N.ptr(eps), N.ptr(lpxz), B, K, D,
                                            N.ptr(z), N.ptr(lq), N.ptr(lp), N.ptr(iw), N.stream_ptr(dev)))
    return {"latent_z": z, "log_q_z_x": lq, "log_p_z": lp, "iw_ll": iw}
