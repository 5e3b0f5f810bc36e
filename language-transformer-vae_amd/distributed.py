"""Row-sharded multi-GPU latent statistics over torch.distributed (RCCL on MI355X, gloo in CPU tests).

The reference's only multi-GPU pattern for this path is shard -> independent compute -> gather through
pickles, with no collective (iw_log_likelihood_x_obs_x_gen/dist_iw_log_likelihood_x_obs_x_gen.py:48-73,
:147).  Here each rank owns a contiguous block of posteriors (mu, logvar) and the z samples drawn from
them.  One exchange step:

  1. all-gather of the packed [n_r, 2D] posterior parameters -> every rank holds all N posteriors;
  2. every rank evaluates log q(z_i) for ITS rows against all N posteriors -- the logsumexp over the
     posterior axis completes locally, so no partial logsumexp is exchanged;
  3. all-reduce (sum) of a handful of float64 partial sums -> marginal KL / MI / TC on every rank.

Ragged shards are supported (shard sizes are exchanged first and the gather is padded).

``local_stats`` is the per-rank compute; the default runs the HIP kernels and raises without a GPU.
The CPU tests inject an oracle-backed callable to exercise the sharding and collective logic.
"""
import math

import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous, balanced split of ``n`` rows: the first ``n % world_size`` ranks get one extra."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _host_staged(group):
    """gloo has no device collectives: with that backend GPU tensors are staged through host memory
    (debug / rehearsal only; RCCL works on device memory directly)."""
    return dist.get_backend(group) == "gloo"


def _all_gather_into(out, x, group):
    if x.is_cuda and _host_staged(group):
        tmp = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(tmp, x.cpu(), group=group)
        out.copy_(tmp)
    else:
        dist.all_gather_into_tensor(out, x, group=group)


def _all_reduce(x, op, group, async_op=False):
    """In-place all-reduce; with ``async_op`` returns the work handle (or None where the collective has
    already completed) and the current stream does NOT wait for the collective."""
    if x.is_cuda and _host_staged(group):
        tmp = x.cpu()
        dist.all_reduce(tmp, op=op, group=group)
        x.copy_(tmp)
        return None
    return dist.all_reduce(x, op=op, group=group, async_op=async_op)


def all_gather_rows(x, group=None, counts=None):
    """All-gather a [n_r, C] tensor whose row count may differ per rank -> [sum n_r, C], plus the row
    offset of every rank.  ``counts`` (rows per rank), when the caller knows them, saves the size
    exchange and its host synchronisation."""
    ws = dist.get_world_size(group)
    if counts is None:
        n_local = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
        counts = torch.empty(ws, dtype=torch.int64, device=x.device)
        _all_gather_into(counts, n_local, group)
        counts = counts.tolist()
    else:
        counts = [int(c) for c in counts]
        assert len(counts) == ws and counts[dist.get_rank(group)] == x.shape[0]
    nmax = max(counts)
    if all(c == nmax for c in counts):
        out = torch.empty((ws * nmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _all_gather_into(out, x.contiguous(), group)
    else:
        pad = torch.zeros((nmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[:x.shape[0]] = x
        buf = torch.empty((ws * nmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _all_gather_into(buf, pad, group)
        out = torch.cat([buf[r * nmax:r * nmax + counts[r]] for r in range(ws)], dim=0)
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    return out, offsets


def hip_local_stats(mu_all, logvar_all, z_local, mu_local, logvar_local, log_norm, prod_marginals, path=0):
    """Per-rank compute on the HIP path: float64 sums over this rank's rows of
    [log q(z), log prod-marginals (or 0), log p(z), log q(z|x)] and the row count.  Without prod-marginals this is
    one C call (lvae_latent_stats_f32: on small problems the sums ride on the kernel that merges the partials)."""
    from . import _native as N
    from . import estimators as E
    z, mu, lv = (N.as_f32(t, n) for t, n in ((z_local, "z"), (mu_local, "mu"), (logvar_local, "logvar")))
    dev = z.device
    if z.dim() != 2 or mu.shape != z.shape or lv.shape != z.shape:
        raise ValueError(f"local rows must share one [n_r, D] shape: z {tuple(z.shape)}, mu {tuple(mu.shape)}, "
                         f"logvar {tuple(lv.shape)}")
    if mu_all.dim() != 2 or mu_all.shape != logvar_all.shape or mu_all.shape[1] != z.shape[1]:
        raise ValueError(f"gathered posteriors must be [N, D] with the local D: mu_all {tuple(mu_all.shape)}, "
                         f"logvar_all {tuple(logvar_all.shape)}, z {tuple(z.shape)}")
    if not prod_marginals:
        mu_a, lv_a = N.as_f32(mu_all, "mu"), N.as_f32(logvar_all, "logvar")
        M, D = z.shape
        Nn = mu_a.shape[0]
        with torch.cuda.device(dev):
            ws, nbytes = E._pair_workspace(M, Nn, D, False, dev)
            log_q_z = torch.empty(M, dtype=torch.float32, device=dev)
            sums = torch.empty(5, dtype=torch.float64, device=dev)
            N.check(N.lib().lvae_latent_stats_f32(N.ptr(z), N.ptr(mu_a), N.ptr(lv_a), M, Nn, D, float(log_norm), path,
                                                  N.ptr(mu), N.ptr(lv), N.ptr(log_q_z), N.ptr(sums), N.ptr(ws), nbytes,
                                                  N.stream_ptr(dev)))
        return sums, log_q_z
    log_q_z, log_q_z_pm = E.log_qz_raw(mu_all, logvar_all, z_local, log_norm, prod_marginals, path=path)
    with torch.cuda.device(dev):
        sums = torch.empty(5, dtype=torch.float64, device=dev)
        N.check(N.lib().lvae_latent_sums_f32(N.ptr(z), N.ptr(mu), N.ptr(lv), N.ptr(log_q_z),
                                             N.ptr(log_q_z_pm), z.shape[0], z.shape[1], N.ptr(sums),
                                             N.stream_ptr(dev)))
    return sums, log_q_z


class HipBlockKernels:
    """The three per-rank device steps of the overlapped exchange on the HIP path (the CPU tests inject an
    oracle-backed object with the same three methods to exercise the collective logic)."""

    @staticmethod
    def partial(mu, logvar, z):
        from . import estimators as E
        return E.log_qz_partial(mu, logvar, z)

    @staticmethod
    def combine(own, rest, log_norm):
        from . import estimators as E
        return E.lse_combine(own, rest, log_norm)

    @staticmethod
    def sums(z_local, mu_local, logvar_local, log_q_z):
        from . import _native as N
        z, mu, lv = (N.as_f32(t, n) for t, n in ((z_local, "z"), (mu_local, "mu"), (logvar_local, "logvar")))
        dev = z.device
        with torch.cuda.device(dev):
            sums = torch.empty(5, dtype=torch.float64, device=dev)
            N.check(N.lib().lvae_latent_sums_f32(N.ptr(z), N.ptr(mu), N.ptr(lv), N.ptr(log_q_z), None, z.shape[0],
                                                 z.shape[1], N.ptr(sums), N.stream_ptr(dev)))
        return sums


class Timeline:
    """Per-phase stamps of one exchange step: HIP events on the current stream for GPU tensors (the stream's own
    clock: what the device did, not when the host queued it), ``time.perf_counter`` for the CPU rehearsal.
    ``phases_ms()`` synchronises and returns {phase: milliseconds between its stamp and the previous one}."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.device = device
        self.marks = []

    def mark(self, name):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.marks.append((name, ev))
        else:
            import time
            self.marks.append((name, time.perf_counter()))

    def phases_ms(self):
        if self.cuda:
            torch.cuda.synchronize(self.device)
        out = {}
        for (_, a), (name, b) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + (a.elapsed_time(b) if self.cuda else (b - a) * 1e3)
        return out


def overlapped_stats(mu_local, logvar_local, z_local, group, log_norm_of, kernels=HipBlockKernels, timeline=None):
    """The exchange step with the all-gather hidden behind compute (equal shard sizes, joint density only).
    The logsumexp over posteriors is order-free, so it is evaluated in blocks:
      1. the all-gathers of mu and logvar ([n_r, D] each -> [world * n_r, D]: two collectives, so that every rank's
         block AND every range of ranks is a contiguous [rows, D] matrix the kernels read in place) are started
         asynchronously (RCCL runs them on its own stream; with the host-staged gloo rehearsal of GPU tensors they
         simply run first);
      2. meanwhile this rank's samples are contracted against its OWN posteriors (1/world of the pairs) into a
         partial logsumexp (lvae_log_qz_partial_f32);
      3. the current stream waits for the collectives; the same samples go against the posteriors of the ranks BEFORE
         this one and of the ranks AFTER it -- two row ranges of the gathered buffers, read where they lie (no
         copy: the first and the last rank have one range) -- and the partials are merged (lvae_lse_combine_f32).
    ``timeline`` (a Timeline): stamps after each phase -- own_block, wait_all_gather (how long the compute stream sat
    behind the collective once its own block was done: ~0 when the all-gather is hidden), rest_block, merge_and_sums.
    Returns (sums [5] float64, log_q_z of the local rows, n_total)."""
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    n_r, D = mu_local.shape
    mu_local, logvar_local = mu_local.contiguous(), logvar_local.contiguous()
    mu_all = torch.empty((ws * n_r, D), dtype=mu_local.dtype, device=mu_local.device)
    lv_all = torch.empty((ws * n_r, D), dtype=mu_local.dtype, device=mu_local.device)
    works = []
    if timeline is not None:
        timeline.mark("start")
    if mu_local.is_cuda and _host_staged(group):
        _all_gather_into(mu_all, mu_local, group)
        _all_gather_into(lv_all, logvar_local, group)
    else:
        works.append(dist.all_gather_into_tensor(mu_all, mu_local, group=group, async_op=True))
        works.append(dist.all_gather_into_tensor(lv_all, logvar_local, group=group, async_op=True))
    acc = kernels.partial(mu_local, logvar_local, z_local)
    if timeline is not None:
        timeline.mark("own_block")
    for w in works:
        w.wait()                         # GPU: the current stream waits for the collective, the host does not
    if timeline is not None:
        timeline.mark("wait_all_gather")
    lo, hi = rank * n_r, (rank + 1) * n_r
    n_total = ws * n_r
    for a, b in ((0, lo), (hi, n_total)):
        if b > a:
            acc = kernels.combine(acc, kernels.partial(mu_all[a:b], lv_all[a:b], z_local), None)
    if timeline is not None:
        timeline.mark("rest_block")
    log_q_z = kernels.combine(acc, None, log_norm_of(n_total))
    sums = kernels.sums(z_local, mu_local, logvar_local, log_q_z)
    if timeline is not None:
        timeline.mark("merge_and_sums")
    return sums, log_q_z, n_total


def sharded_latent_stats(mu_local, logvar_local, z_local, method="all", dataset_size=None,
                         prod_marginals=False, group=None, local_stats=hip_local_stats, counts=None,
                         overlap=None, block_kernels=HipBlockKernels):
    """Marginal KL / MI (/ TC, dimension-wise KL) over posteriors and samples sharded by rows: the device
    part (``sharded_latent_sums``) followed by one host read of the five global sums (``stats_from_sums``).

    mu_local, logvar_local, z_local: this rank's [n_r, D] rows (z_i drawn from posterior i).
    Returns a dict of Python floats identical on every rank, with the definitions of
    loss_and_optimisation.py:816-824: mi = E[log q(z|x)] - E[log q(z)], marginal_kl = E[log q(z)] -
    E[log p(z)], tc = E[log q(z)] - E[log prod q(z_d)], dim_kl = E[log prod q(z_d)] - E[log p(z)];
    plus ``log_q_z_local`` (this rank's per-sample vector, left sharded).  ``counts``: rows per rank if
    known (skips the size exchange).  ``overlap`` (default: whenever it applies -- the HIP path, equal
    shard sizes known through ``counts``, no prod-marginals, more than one rank): hide the all-gather behind
    the local block of the pair matrix, see ``overlapped_stats``."""
    sums, log_q_z, n_total = sharded_latent_sums(mu_local, logvar_local, z_local, method, dataset_size, prod_marginals,
                                                 group, local_stats, counts, overlap, block_kernels)
    return stats_from_sums(sums, n_total, log_q_z, prod_marginals)


def stats_from_sums(sums, n_total, log_q_z_local, prod_marginals=False):
    """The estimates of loss_and_optimisation.py:816-824 from the five global float64 sums (reads them to the host)."""
    s_lq, s_pm, s_lp, s_lqx, cnt = sums.tolist()
    out = {
        "n_total": int(n_total), "m_total": int(round(cnt)),
        "log_q_z": s_lq / cnt, "log_p_z": s_lp / cnt, "log_q_z_x": s_lqx / cnt,
        "marginal_kl": (s_lq - s_lp) / cnt, "mi": (s_lqx - s_lq) / cnt,
        "log_q_z_local": log_q_z_local,
    }
    if prod_marginals:
        out.update(log_q_z_prod_marg=s_pm / cnt, tc=(s_lq - s_pm) / cnt, dim_kl=(s_pm - s_lp) / cnt)
    return out


def sharded_latent_sums(mu_local, logvar_local, z_local, method="all", dataset_size=None,
                        prod_marginals=False, group=None, local_stats=hip_local_stats, counts=None,
                        overlap=None, block_kernels=HipBlockKernels, async_reduce=False, timeline=None):
    """Exchange + per-rank compute + all-reduce of ``sharded_latent_stats`` without its host read: returns
    (sums [5] float64 on the compute device, identical on every rank: sum log q(z), sum log prod-marginals,
    sum log p(z), sum log q(z|x), row count; this rank's log q(z) vector; n_total).  Nothing here waits on
    the host, so consecutive calls queue behind each other on the device.  ``async_reduce``: the compute
    stream does not wait for the final all-reduce either (the next call's kernels start at once); a fourth
    return value is then the collective's work handle (or None), to be ``wait()``-ed before ``sums`` is read.
    ``timeline`` (a Timeline; overlapped exchange only): per-phase stamps, the all-reduce included."""
    if method == "chen":
        assert dataset_size is not None

    def log_norm_of(n_total):
        return math.log(n_total * dataset_size) if method == "chen" else math.log(n_total)

    hip_path = local_stats is hip_local_stats or block_kernels is not HipBlockKernels
    can_overlap = (hip_path and not prod_marginals and counts is not None
                   and dist.get_world_size(group) > 1 and len(set(int(c) for c in counts)) == 1
                   and int(counts[0]) == mu_local.shape[0] and mu_local.shape[0] > 0)
    if overlap is None:
        overlap = can_overlap
    if overlap:
        if not can_overlap:
            raise ValueError("overlap=True needs the HIP path, equal shard sizes given by `counts`, "
                             "prod_marginals=False and world_size > 1")
        sums, log_q_z, n_total = overlapped_stats(mu_local, logvar_local, z_local, group, log_norm_of, block_kernels,
                                                  timeline)
    else:
        packed = torch.cat([mu_local, logvar_local], dim=1).contiguous()
        gathered, _ = all_gather_rows(packed, group, counts)
        D = mu_local.shape[1]
        mu_all, logvar_all = gathered[:, :D].contiguous(), gathered[:, D:].contiguous()
        n_total = mu_all.shape[0]
        sums, log_q_z = local_stats(mu_all, logvar_all, z_local, mu_local, logvar_local, log_norm_of(n_total),
                                    prod_marginals)
    sums = sums.to(torch.float64).contiguous()
    work = _all_reduce(sums, dist.ReduceOp.SUM, group, async_op=async_reduce)
    if timeline is not None and overlap:
        if work is not None:
            work.wait()
        timeline.mark("all_reduce")
    if async_reduce:
        return sums, log_q_z, n_total, work
    return sums, log_q_z, n_total


def hip_reparam_iw(mu, logvar, eps, log_p_x_z):
    """Per-rank compute of sharded_iw_log_p_x on the HIP path: the reduction-only IW kernel -> iw_ll [B_r]."""
    from . import estimators as E
    return E.reparam_iw(mu, logvar, eps, log_p_x_z, return_z=False, return_densities=False)["iw_ll"]


def sharded_iw_log_p_x(mu_local, logvar_local, eps_local, log_p_x_z_local, group=None, gather=True, counts=None,
                       reparam_iw=hip_reparam_iw, timeline=None):
    """IW log-likelihood with sentences sharded over ranks (independent per sentence, like the
    DistributedSampler split at dist_iw_...py:86-87); the optional all-gather of the [B_r] results
    replaces the reference's pickle concatenation (:48-73).  ``counts``: sentences per rank if known (saves the size
    exchange and its host synchronisation); ``timeline``: stamps after the kernel and after the gather."""
    if timeline is not None:
        timeline.mark("start")
    iw = reparam_iw(mu_local, logvar_local, eps_local, log_p_x_z_local)
    if timeline is not None:
        timeline.mark("kernel")
    if not gather:
        return iw
    out, _ = all_gather_rows(iw.reshape(-1, 1), group, counts)
    if timeline is not None:
        timeline.mark("all_gather")
    return out.reshape(-1)


def hip_kernel_sum(a, b, alphas, dist_eps, exclude_diag, diag_offset, symmetric=False):
    """Per-rank compute of sharded_mmd on the HIP path: float64 device scalar."""
    from . import estimators as E
    return E.kernel_sum(a, b, alphas, dist_eps, exclude_diag, diag_offset, symmetric=symmetric)


def sharded_mmd(x_local, y_local, alphas, dist_eps=0.0, unbiased=False, group=None,
                kernel_sum=hip_kernel_sum):
    """Gaussian-kernel MMD between two samples whose rows are sharded over ranks (the samples
    are replicated by all-gather, the kernel-matrix rows are partitioned, three float64 block sums are
    all-reduced).  After the gather the compute partition is free: for the two self-terms every rank
    evaluates two mirrored column chunks of the upper triangle (``mirrored_chunks``: half the work,
    balanced over ranks); for the cross term it keeps its own rows of x against all of y.
    ``unbiased=True``, ``dist_eps=1e-5`` is the torch-two-sample statistic used at
    loss_and_optimisation.py:617-619 (diagonals excluded, 1/(n(n-1)) weights); ``unbiased=False``,
    ``dist_eps=0``, ``alphas=[1/D^2]`` is maximum_mean_discrepancy (:46-60).  Returns a dict of Python
    floats, identical on every rank."""
    x_all, xoff = all_gather_rows(x_local.contiguous(), group)
    y_all, yoff = all_gather_rows(y_local.contiguous(), group)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    parts = []
    for full in (x_all, y_all):
        acc = torch.zeros((), dtype=torch.float64, device=full.device)
        for lo, hi in mirrored_chunks(full.shape[0], world, rank):
            acc = acc + kernel_sum(full[lo:hi].contiguous(), full, alphas, dist_eps, unbiased, lo, True).reshape(())
        parts.append(acc)
    parts.append(kernel_sum(x_local, y_all, alphas, dist_eps, False, 0, False).reshape(())
                 if x_local.shape[0] else torch.zeros((), dtype=torch.float64, device=x_all.device))
    sums = torch.stack(parts).to(torch.float64)
    _all_reduce(sums, dist.ReduceOp.SUM, group)
    sxx, syy, sxy = sums.tolist()
    n1, n2 = float(x_all.shape[0]), float(y_all.shape[0])
    c11 = n1 * (n1 - 1.0) if unbiased else n1 * n1
    c22 = n2 * (n2 - 1.0) if unbiased else n2 * n2
    return {"mmd": sxx / c11 + syy / c22 - 2.0 * sxy / (n1 * n2), "sum_kxx": sxx, "sum_kyy": syy,
            "sum_kxy": sxy, "n1": int(n1), "n2": int(n2)}


def hip_permutation_sums(cols, col_offset, pooled, labels, alphas, dist_eps, symmetric):
    from . import estimators as E
    return E.mmd_permutation_sums(cols, col_offset, pooled, labels, alphas, dist_eps, symmetric)


def mirrored_chunks(n, world_size, rank, align=32):
    """Column ranges of the pooled sample for one rank when only the upper triangle of the kernel matrix is
    evaluated: the columns are cut into 2*world_size chunks (multiples of ``align``) and rank r takes chunks
    r and 2*world_size-1-r, whose work (rows beyond the column) adds up to the same for every rank."""
    nchunks = 2 * world_size
    cs = -(-n // nchunks)
    cs = -(-cs // align) * align
    out = []
    for k in (rank, nchunks - 1 - rank):
        lo, hi = min(k * cs, n), min((k + 1) * cs, n)
        if hi > lo:
            out.append((lo, hi))
    return out


def sharded_mmd_permutation_test(x_local, y_local, alphas, labels, dist_eps=1e-5, group=None,
                                 permutation_sums=hip_permutation_sums, counts=None, timeline=None):
    """MMD permutation test with both samples sharded by rows (samples replicated by
    all-gather, kernel-matrix columns partitioned, all-reduce of the [P, 2] label sums and the total).
    After the gather every rank holds the pooled sample, so the COMPUTE partition is free: each rank
    evaluates two mirrored column chunks of the upper triangle (``mirrored_chunks``), half the work of the
    full matrix and balanced over ranks.
    ``labels``: [P, n1 + n2] uint8, identical on every rank (row 0 = observed labeling: first n1 ones), in
    POOLED order = all of x (rank order) followed by all of y.  ``counts`` = (rows of x per rank, rows of y per
    rank) if known (saves the two size exchanges); ``timeline``: stamps after the gathers, the kernels and the
    all-reduce.  Returns a dict identical on every rank."""
    if timeline is not None:
        timeline.mark("start")
    x_all, _ = all_gather_rows(x_local.contiguous(), group, None if counts is None else counts[0])
    y_all, _ = all_gather_rows(y_local.contiguous(), group, None if counts is None else counts[1])
    if timeline is not None:
        timeline.mark("all_gather")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n1, n2 = x_all.shape[0], y_all.shape[0]
    pooled = torch.cat((x_all, y_all), 0).contiguous()
    P = labels.shape[0]
    sums = torch.zeros(2 * P + 1, dtype=torch.float64, device=pooled.device)
    for lo, hi in mirrored_chunks(n1 + n2, world, rank):
        q, lr, tot = permutation_sums(pooled[lo:hi].contiguous(), lo, pooled, labels, alphas, dist_eps, True)
        sums += torch.cat((q.reshape(-1), lr.reshape(-1), tot.reshape(1))).to(torch.float64)
    if timeline is not None:
        timeline.mark("kernels")
    _all_reduce(sums, dist.ReduceOp.SUM, group)
    if timeline is not None:
        timeline.mark("all_reduce")
    s = sums.cpu()
    q, lr, tot = s[:P], s[P:2 * P], s[2 * P]
    k12 = lr - q
    k22 = tot - 2.0 * lr + q
    stats = q / (n1 * (n1 - 1.0)) + k22 / (n2 * (n2 - 1.0)) - 2.0 * k12 / (float(n1) * float(n2))
    return {"statistic": float(stats[0]), "statistics": stats[1:].tolist(),
            "pval": float((stats[1:] >= stats[0]).double().mean()) if P > 1 else float("nan")}


class HipLseKernels:
    """Device steps of ``combine_partial_lse`` on the HIP path (lvae_lse_rebase_f32, lvae_lse_combine_f32); the CPU
    tests inject an oracle-backed object with the same two methods."""

    @staticmethod
    def rebase(row_max, row_sum, new_max):
        from . import _native as N
        m, s, g = (N.as_f32(t, "partial") for t in (row_max, row_sum, new_max))
        dev = m.device
        with torch.cuda.device(dev):
            out = torch.empty_like(s)
            N.check(N.lib().lvae_lse_rebase_f32(N.ptr(m), N.ptr(s), N.ptr(g), m.numel(), N.ptr(out),
                                                N.stream_ptr(dev)))
        return out

    @staticmethod
    def finish(gmax, gsum, log_norm):
        from . import estimators as E
        return E.lse_combine((gmax, gsum), None, log_norm)


def combine_partial_lse(row_max, row_sum, log_norm, group=None, kernels=HipLseKernels):
    """Column-sharded alternative: every rank holds ALL z rows but only its posteriors and has produced
    (row_max, row_sum) with lvae_log_qz_partial_f32.  all-reduce(max); the sums re-expressed against the common
    maximum on the device (lvae_lse_rebase_f32); all-reduce(sum); log q(z) finished on the device
    (lvae_lse_combine_f32)."""
    gmax = row_max.clone()
    _all_reduce(gmax, dist.ReduceOp.MAX, group)
    scaled = kernels.rebase(row_max, row_sum, gmax)
    _all_reduce(scaled, dist.ReduceOp.SUM, group)
    return kernels.finish(gmax, scaled, log_norm)
