"""Mirror of the hot-path tail of NewsVAE/modules/encoder.py (EncoderNewsVAE static methods and the
part of ``encode`` that follows the transformer forward)."""
import torch

from . import _native as N
from . import estimators as E


class _Reparam(torch.autograd.Function):
    """z = mu + eps * exp(0.5 logvar) with a HIP backward (lvae_reparam_bwd_f32)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu_d, lv_d = N.as_f32(mu.detach(), "mu"), N.as_f32(logvar.detach(), "logvar")
        eps = N.as_f32(eps.detach(), "eps")
        B, K, D = eps.shape
        dev = mu_d.device
        with torch.cuda.device(dev):
            z = torch.empty_like(eps)
            N.check(N.lib().lvae_reparam_iw_f32(N.ptr(mu_d), N.ptr(lv_d), N.ptr(eps), None, B, K, D, N.ptr(z), None,
                                                None, None, N.stream_ptr(dev)))
        ctx.save_for_backward(lv_d, eps)
        return z

    @staticmethod
    def backward(ctx, g):
        logvar, eps = ctx.saved_tensors
        B, K, D = eps.shape
        dev = eps.device
        g = N.as_f32(g.detach(), "grad").expand(eps.shape).contiguous()
        with torch.cuda.device(dev):
            d_mu = torch.empty_like(logvar) if ctx.needs_input_grad[0] else None
            d_lv = torch.empty_like(logvar) if ctx.needs_input_grad[1] else None
            if d_mu is not None or d_lv is not None:
                N.check(N.lib().lvae_reparam_bwd_f32(N.ptr(logvar), N.ptr(eps), N.ptr(g), B, K, D, N.ptr(d_mu),
                                                     N.ptr(d_lv), N.stream_ptr(dev)))
        return d_mu, d_lv, None


_TAIL_MODES = {None: 0, "auto": 0, True: 0, "one_wg": 1, "tiny": 1, "multi_wg": 2, "mid": 2}


def _tail_workspace(B, K, D, dev):
    nbytes = N._i64(0)
    N.check(N.lib().lvae_encode_tail_workspace_bytes(int(B), int(K), int(D), nbytes))
    return N.workspace(max(nbytes.value, 256), dev), nbytes.value


def _tiny_forward(mu, logvar, eps, prior_sample, log_norm, want_z=True, mode=0):
    """The fused tail (lvae_encode_tail_ws_f32: one workgroup up to B = 32, B / 4 workgroups beyond): z, log q(z|x),
    log p(z), and the scalars [mean log q(z), mean prod-marginals, mmd]."""
    B, K, D = eps.shape
    dev = mu.device
    with torch.cuda.device(dev):
        z = torch.empty_like(eps) if want_z else None
        lqx = torch.empty(B, K, dtype=torch.float32, device=dev)
        lpz = torch.empty(B, K, dtype=torch.float32, device=dev)
        scal = torch.empty(3, dtype=torch.float32, device=dev)
        ws, nbytes = _tail_workspace(B, K, D, dev)
        N.check(N.lib().lvae_encode_tail_ws_f32(N.ptr(mu), N.ptr(logvar), N.ptr(eps), N.ptr(prior_sample), B, K, D,
                                                float(log_norm), N.ptr(z), N.ptr(lqx), N.ptr(lpz), None, None,
                                                N.ptr(scal), mode, N.ptr(ws), nbytes, N.stream_ptr(dev)))
    return z, lqx, lpz, scal


class _EncodeTailTiny(torch.autograd.Function):
    """The whole differentiable tail of encode() for one small batch: one HIP launch forward
    (lvae_encode_tail_f32), one backward (lvae_encode_tail_bwd_f32)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps, prior_sample, log_norm, mode=0):
        mu_d, lv_d = N.as_f32(mu.detach(), "mu"), N.as_f32(logvar.detach(), "logvar")
        eps_d, pr_d = N.as_f32(eps.detach(), "eps"), N.as_f32(prior_sample.detach(), "prior_sample")
        z, lqx, lpz, scal = _tiny_forward(mu_d, lv_d, eps_d, pr_d, log_norm, mode=mode)
        ctx.save_for_backward(mu_d, lv_d, eps_d)
        ctx.log_norm = float(log_norm)
        ctx.mode = mode
        lq, pm, mmd = scal[0], scal[1], scal[2]
        ctx.mark_non_differentiable(mmd)              # quirk Q5: the reference's kernel detaches
        return z, lqx, lpz, lq, pm, mmd

    @staticmethod
    def backward(ctx, g_z, g_lqx, g_lpz, g_lq, g_pm, _g_mmd):
        mu, logvar, eps = ctx.saved_tensors
        B, K, D = eps.shape
        dev = mu.device
        fix = lambda g, shape: None if g is None else N.as_f32(g.detach(), "grad").expand(shape).contiguous()
        g_z, g_lqx, g_lpz = fix(g_z, (B, K, D)), fix(g_lqx, (B, K)), fix(g_lpz, (B, K))
        g_lq, g_pm = fix(g_lq, (1,)), fix(g_pm, (1,))
        with torch.cuda.device(dev):
            d_mu = torch.empty_like(mu) if ctx.needs_input_grad[0] else None
            d_lv = torch.empty_like(logvar) if ctx.needs_input_grad[1] else None
            if d_mu is not None or d_lv is not None:
                ws, nbytes = _tail_workspace(B, K, D, dev)
                N.check(N.lib().lvae_encode_tail_bwd_ws_f32(N.ptr(mu), N.ptr(logvar), N.ptr(eps), B, K, D, ctx.log_norm,
                                                            N.ptr(g_z), N.ptr(g_lqx), N.ptr(g_lpz), N.ptr(g_lq),
                                                            N.ptr(g_pm), N.ptr(d_mu), N.ptr(d_lv), ctx.mode, N.ptr(ws),
                                                            nbytes, N.stream_ptr(dev)))
        return d_mu, d_lv, None, None, None, None


def tiny_tail_supported(B, K, D):
    """Whether the one-workgroup tail (csrc/encode_tail_tiny.hip) serves this batch: B <= 128, B * D <= 4096, ..."""
    return bool(N.lib().lvae_encode_tail_supported(int(B), int(K), int(D)))


def fused_tail_supported(B, K, D):
    """Whether a fused tail serves this batch (one workgroup, or csrc/encode_tail_mid.hip: D <= 64, B <= 4096)."""
    return bool(N.lib().lvae_encode_tail_ws_supported(int(B), int(K), int(D)))


# Above this batch size the general kernels (matrix cores) take over from the fused multi-workgroup tail, whose pair
# loops are plain vector code (tools/bench_train_tail.py)
FUSED_TAIL_MAX_B = 1024


def reparameterize(mu, logvar, n_samples=1, eps=None):
    """EncoderNewsVAE.reparameterize (modules/encoder.py:130-159): z[b,k,:] = mu_b + eps_bk * exp(0.5 *
    logvar_b) -> [batch, n_samples, nz].  The reference draws eps as one contiguous [B,K,D] normal_()
    (:156); pass ``eps`` to inject it."""
    batch_size, nz = mu.size()
    if eps is None:
        eps = torch.zeros(batch_size, n_samples, nz, dtype=torch.float32, device=mu.device).normal_()
    if tuple(eps.shape) != (batch_size, n_samples, nz):
        raise ValueError("eps must be [batch, n_samples, nz]")
    if E._needs_grad(mu, logvar):
        return _Reparam.apply(mu, logvar, eps)
    mu, logvar, eps = N.as_f32(mu, "mu"), N.as_f32(logvar, "logvar"), N.as_f32(eps, "eps")
    dev = mu.device
    with torch.cuda.device(dev):
        z = torch.empty_like(eps)
        N.check(N.lib().lvae_reparam_iw_f32(N.ptr(mu), N.ptr(logvar), N.ptr(eps), None, batch_size, n_samples,
                                            nz, N.ptr(z), None, None, None, N.stream_ptr(dev)))
    return z


def sample_from_prior(latent_size=768, n_samples=8, device_name="cuda:0"):
    """EncoderNewsVAE.sample_from_prior (modules/encoder.py:161-179): [n_samples, latent_size] standard
    normal draws (RNG only; no arithmetic on the hot path)."""
    return torch.randn(n_samples, latent_size, device=device_name)


def encode_tail(mu, logvar, n_samples=1, dataset_size=10000, return_log_q_z_x=True, return_log_p_z=True,
                return_log_q_z=True, eps=None, prior_sample=None, fused=None):
    """What EncoderNewsVAE.encode does after the transformer has produced (mu, logvar)
    (modules/encoder.py:79-128): sample, per-sample densities, the Chen et al. minibatch estimate of
    log q(z) / prod-marginals on the first sample (quirk Q4: dataset_size defaults to 10000, :68), and
    the biased MMD of the first sample against the prior.
    ``fused`` (default: the training-time batch sizes, B <= 1024 at D <= 64): everything in one or two launches, and
    as many for the backward (lvae_encode_tail_ws_f32 / _bwd_ws_f32: one workgroup up to B = 32, B / 4 workgroups
    beyond); "one_wg" / "multi_wg" force a form; ``fused=False`` forces the general kernels."""
    import math
    B, D = mu.shape
    if eps is None:
        eps = torch.zeros(B, n_samples, D, dtype=torch.float32, device=mu.device).normal_()
    if fused is None:
        fused = mu.is_cuda and B <= FUSED_TAIL_MAX_B and fused_tail_supported(B, eps.shape[1], D)
    mode = _TAIL_MODES[fused] if fused is not False else 0
    if fused:
        if prior_sample is None:
            prior_sample = torch.randn(B, D, dtype=torch.float32, device=mu.device)        # :53, randn_like(latent_z)
        log_norm = math.log(B * dataset_size)                                                 # "chen", :207
        if E._needs_grad(mu, logvar):
            latent_z, log_q_z_x, log_p_z, log_q_z, log_q_z_prod_marg, mmd_loss = _EncodeTailTiny.apply(
                mu, logvar, eps, prior_sample, log_norm, mode)
        else:
            mu, logvar = N.as_f32(mu, "mu"), N.as_f32(logvar, "logvar")
            latent_z, log_q_z_x, log_p_z, scal = _tiny_forward(mu, logvar, N.as_f32(eps, "eps"),
                                                               N.as_f32(prior_sample, "prior_sample"), log_norm, mode=mode)
            log_q_z, log_q_z_prod_marg, mmd_loss = scal[0], scal[1], scal[2]
        return {"mu": mu, "logvar": logvar, "latent_z": latent_z, "mmd_loss": mmd_loss,
                "log_q_z": log_q_z if return_log_q_z else None,
                "log_q_z_prod_marg": log_q_z_prod_marg if return_log_q_z else None,
                "log_q_z_x": log_q_z_x if return_log_q_z_x else None,
                "log_p_z": log_p_z if return_log_p_z else None}
    if E._needs_grad(mu, logvar):
        # training-time call: the same sequence of differentiable estimators as the reference (:80-107),
        # every one with a HIP forward and a HIP backward
        latent_z = reparameterize(mu, logvar, n_samples, eps)
        single_sample_z = latent_z[:, 0, :]
        log_q_z_x = log_p_z = log_q_z = log_q_z_prod_marg = None
        if return_log_q_z_x:
            log_q_z_x = E.sample_log_likelihood(latent_z, mu=mu, logvar=logvar, reduce_latent_dim=True,
                                                reduce_batch_dim=False)
        if return_log_p_z:
            log_p_z = E.sample_log_likelihood(latent_z, reduce_latent_dim=True, reduce_batch_dim=False)
        if return_log_q_z:
            log_q_z, log_q_z_prod_marg = E.approximate_log_q_z(mu, logvar, single_sample_z, method="chen",
                                                               dataset_size=dataset_size, prod_marginals=True)
        mmd_loss = E.maximum_mean_discrepancy(single_sample_z, prior_sample)
        return {"mu": mu, "logvar": logvar, "latent_z": latent_z, "mmd_loss": mmd_loss, "log_q_z": log_q_z,
                "log_q_z_prod_marg": log_q_z_prod_marg, "log_q_z_x": log_q_z_x, "log_p_z": log_p_z}
    mu, logvar = N.as_f32(mu, "mu"), N.as_f32(logvar, "logvar")
    fused = E.reparam_iw(mu, logvar, eps, None, return_z=True)
    latent_z = fused["latent_z"]
    single_sample_z = latent_z[:, 0, :].contiguous()                                   # :81
    log_q_z = log_q_z_prod_marg = None
    if return_log_q_z:                                                                 # :100-104
        log_q_z, log_q_z_prod_marg = E.approximate_log_q_z(mu, logvar, single_sample_z, method="chen",
                                                           dataset_size=dataset_size, prod_marginals=True)
    mmd_loss = E.maximum_mean_discrepancy(single_sample_z, prior_sample)               # :107
    return {"mu": mu, "logvar": logvar, "latent_z": latent_z, "mmd_loss": mmd_loss, "log_q_z": log_q_z,
            "log_q_z_prod_marg": log_q_z_prod_marg,
            "log_q_z_x": fused["log_q_z_x"] if return_log_q_z_x else None,
            "log_p_z": fused["log_p_z"] if return_log_p_z else None}
