"""Pareto-efficient checkpoint selection of NewsVAE/utils_train.py (:416-516): the host bookkeeping as in the
reference, with the one data-sized step -- the Kolmogorov-Smirnov distance between the two importance-weighted
log-likelihood vectors (:496-497, scipy.stats.ks_2samp) -- on the GPU."""
import math

import numpy as np
import torch

from . import _native as N


def ks_2samp(data1, data2):
    """(statistic, pvalue) of the two-sided two-sample KS test; ``statistic`` is what scipy.stats.ks_2samp
    returns for the same data (utils_train.py:496-497; bit-identical: integer counts, fp64 division).
    ``pvalue`` is the Smirnov limiting form Q_KS(sqrt(n1 n2 / (n1 + n2)) D) -- the reference discards it
    (:498); scipy switches between an exact and an asymptotic computation by sample size."""
    a, b = (torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t) for t in (data1, data2))
    if not a.is_cuda and torch.cuda.is_available():
        a, b = a.cuda(), b.cuda()
    a, b = N.as_f32(a.reshape(-1), "data1"), N.as_f32(b.reshape(-1), "data2")
    n1, n2 = a.numel(), b.numel()
    dev = a.device
    with torch.cuda.device(dev):
        nbytes = N._i64(0)
        N.check(N.lib().lvae_ks_2samp_workspace_bytes(n1, n2, nbytes))
        ws = N.workspace(nbytes.value, dev)
        out = torch.empty(3, dtype=torch.float64, device=dev)
        N.check(N.lib().lvae_ks_2samp_f32(N.ptr(a), n1, N.ptr(b), n2, N.ptr(out), N.ptr(ws), nbytes.value,
                                          N.stream_ptr(dev)))
    d = float(out[0])
    if math.isnan(d):                 # a NaN in either sample: (nan, nan), as scipy's nan_policy="propagate"
        return d, d
    lam = math.sqrt(n1 * n2 / (n1 + n2)) * d
    if lam < 1e-3:
        p = 1.0
    else:
        p = 2.0 * sum((-1) ** (k - 1) * math.exp(-2.0 * k * k * lam * lam) for k in range(1, 101))
    return d, min(max(p, 0.0), 1.0)


def is_pareto_efficient_simple(costs):
    """utils_train.py:416-431 (the reference's variant for GAINS: higher is better).  Boolean [n_points]
    array; a point is dropped iff some other point is >= it in every coordinate, and of identical points
    the first is kept (the sequential sweep of the reference has exactly this fixed point)."""
    g = np.asarray(costs, dtype=np.float64)
    ge = (g[None, :, :] >= g[:, None, :]).all(axis=2)            # ge[i, j]: j >= i everywhere
    gt = (g[None, :, :] > g[:, None, :]).any(axis=2)             # gt[i, j]: j > i somewhere
    n = g.shape[0]
    earlier = np.arange(n)[None, :] < np.arange(n)[:, None]      # j < i
    dominated = ge & (gt | earlier)
    np.fill_diagonal(dominated, False)
    return ~dominated.any(axis=1)


PARETO_KEYS = (("pareto_check_use_rate", "kl_analytical"), ("pareto_check_use_iw_ll_p_w", "iw_ll_p_w"),
               ("pareto_check_use_iw_ll_x_gen_p_w", "iw_ll_x_gen_p_w"), ("pareto_check_use_D_ks", "-D_ks"),
               ("pareto_check_use_distortion", "-distortion"), ("pareto_check_use_tts_mmd", "-tts_mmd"),
               ("pareto_check_use_elbo", "elbo"))


def prepare_pareto_dict(config):
    """utils_train.py:434-461: one empty list per metric switched on in ``config`` (same keys, same order)."""
    return {key: [] for flag, key in PARETO_KEYS if getattr(config, flag, False)}


def determine_pareto_checkpoint(val_epoch_stats, epoch_pareto_effiency_dict, epoch, logging=True,
                                decoder_only=False):
    """utils_train.py:464-516: append this epoch's summary of every tracked metric and return the dict and
    the list of Pareto-efficient epochs.  "-D_ks" is the negated KS distance between the epoch's two IW-LL
    vectors (:496-499; computed on the GPU) -- and, as in the reference, it REPLACES the list by a scalar
    instead of being appended (:499), so the point matrix at :502 cannot be built and the call raises
    IndexError, there and here (train.py:228: "not using D_ks for pareto checkpointing anymore"); use
    ``ks_2samp`` directly.
    ``logging`` (default True as in the reference, :464): the record {"pareto epoch": epoch, "pareto <metric>":
    this epoch's value} of :510-514 goes to ``wandb.log`` when wandb is importable (the reference's sink; it is
    not part of this image), otherwise to the ``lvae.pareto`` Python logger; a callable receives the record
    itself.  "Efficient epochs: [...]" is printed as at :508."""
    metrics = list(epoch_pareto_effiency_dict.keys())
    for metric in metrics:
        if metric in ("kl_analytical", "iw_ll_p_w", "iw_ll_x_gen_p_w", "elbo"):
            epoch_pareto_effiency_dict[metric].append(np.mean(val_epoch_stats[metric]))
        elif metric == "-distortion":
            epoch_pareto_effiency_dict[metric].append(-np.mean(val_epoch_stats["reconstruction_loss"]))
        elif metric == "-tts_mmd":
            epoch_pareto_effiency_dict[metric].append(-np.mean(val_epoch_stats["tts_mmd_loss"]))
        elif metric == "-D_ks":
            d, _ = ks_2samp(val_epoch_stats["iw_ll_p_w"], val_epoch_stats["iw_ll_x_gen_p_w"])
            epoch_pareto_effiency_dict[metric] = np.float64(-d)
    points = np.asarray([[epoch_pareto_effiency_dict[m][i] for m in metrics] for i in range(epoch + 1)])
    efficient_epochs = np.where(is_pareto_efficient_simple(points))[0].tolist()
    print("Efficient epochs:", efficient_epochs)
    if logging:
        record = {"pareto epoch": epoch}
        for m in metrics:
            record[f"pareto {m}"] = epoch_pareto_effiency_dict[m][-1]
        if callable(logging):
            logging(record)
        else:
            try:
                import wandb
                wandb.log(record)
            except ImportError:
                import logging as _pylog
                _pylog.getLogger("lvae.pareto").info("%s", record)
    return epoch_pareto_effiency_dict, efficient_epochs
