"""Host-side mirror of the reference's latent estimators (NewsVAE/loss_and_optimisation.py).

Same names, positional order, defaults and return shapes as the reference functions cited on each
definition; the bodies call the HIP kernels through the C ABI (``_native``).  Forward-only: inputs that
require grad are rejected.
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
    log_q_z, log_q_z_prod_marg = log_qz_raw(mu, logvar, latent_z, log_norm, prod_marginals, path=path)
    if reduce_mean:
        log_q_z = log_q_z.mean()
        log_q_z_prod_marg = log_q_z_prod_marg.mean()     # Q1: None.mean() when prod_marginals=False
    return log_q_z, log_q_z_prod_marg
