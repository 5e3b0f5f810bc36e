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
                return_log_q_z=True, eps=None, prior_sample=None):
    """What EncoderNewsVAE.encode does after the transformer has produced (mu, logvar)
    (modules/encoder.py:79-128): sample, per-sample densities, the Chen et al. minibatch estimate of
    log q(z) / prod-marginals on the first sample (quirk Q4: dataset_size defaults to 10000, :68), and
    the biased MMD of the first sample against the prior."""
    B, D = mu.shape
    if eps is None:
        eps = torch.zeros(B, n_samples, D, dtype=torch.float32, device=mu.device).normal_()
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
