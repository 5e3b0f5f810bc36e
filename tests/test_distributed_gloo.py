"""world_size-2 CPU (gloo) tests of the multi-GPU layer: row sharding, the padded all-gather of ragged
shards, the all-reduce of float64 partial sums, the partial-logsumexp combine and the sharded IW gather.
The per-rank compute is injected (NumPy oracle) because the HIP kernels need a GPU; what is under test
here is the partition / collective / combine logic, which is backend independent."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_stats(mu_all, logvar_all, z_local, mu_local, logvar_local, log_norm, prod_marginals):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle_np as onp
    mu_a, lv_a, z = (t.numpy().astype(np.float64) for t in (mu_all, logvar_all, z_local))
    inv = np.exp(-lv_a)
    dens = -0.5 * (lv_a[None] + onp.LOG_2PI + (z[:, None, :] - mu_a[None]) ** 2 * inv[None])
    lq = onp._logsumexp(dens.sum(2), 1) - log_norm
    pm = (onp._logsumexp(dens, 1) - log_norm).sum(1) if prod_marginals else np.zeros(len(z))
    lp = onp.sample_log_likelihood(z, reduce_batch_dim=False)
    lqx = onp.sample_log_likelihood(z, mu_local.numpy(), logvar_local.numpy(), reduce_batch_dim=False)
    sums = torch.tensor([lq.sum(), pm.sum(), lp.sum(), lqx.sum(), float(len(z))], dtype=torch.float64)
    return sums, torch.from_numpy(lq)


class _OracleLseKernels:
    """CPU stand-in for distributed.HipLseKernels (float64 arithmetic, float32 storage like the device steps)."""

    @staticmethod
    def rebase(row_max, row_sum, new_max):
        return (row_sum.double() * torch.exp(row_max.double() - new_max.double())).float()

    @staticmethod
    def finish(gmax, gsum, log_norm):
        return (gmax.double() + torch.log(gsum.double()) - log_norm).float()


def _worker(rank, world, port, n, D, ragged, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvae import distributed as LD
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((n, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((n, D)) - 1).astype(np.float32)
    z = (mu + rng.standard_normal((n, D)).astype(np.float32) * np.exp(0.5 * lv)).astype(np.float32)
    if ragged:
        lo, hi = (0, n // 3) if rank == 0 else (n // 3, n)
    else:
        lo, hi = LD.shard_bounds(n, world, rank)
    t = lambda a: torch.from_numpy(a[lo:hi].copy())
    out = LD.sharded_latent_stats(t(mu), t(lv), t(z), method="chen", dataset_size=1000, prod_marginals=True,
                                  local_stats=_oracle_local_stats)
    # partial-logsumexp combine: every rank holds all z rows but only its posterior columns
    from oracle import oracle_np as onp
    dens = onp.log_prob_pairs(z, mu[lo:hi], np.exp(lv[lo:hi].astype(np.float64)))     # [cols, rows]
    rmax = dens.max(0)
    rsum = np.exp(dens - rmax[None]).sum(0)
    comb = LD.combine_partial_lse(torch.from_numpy(rmax).float(), torch.from_numpy(rsum).float(), math.log(n),
                                  kernels=_OracleLseKernels)
    # ragged all-gather
    gathered, offs = LD.all_gather_rows(torch.from_numpy(mu[lo:hi].copy()))
    ok_gather = bool(torch.equal(gathered, torch.from_numpy(mu))) and offs[-1] == n
    res = {k: v for k, v in out.items() if not isinstance(v, torch.Tensor)}
    q.put((rank, res, comb.numpy(), ok_gather, out["log_q_z_local"].numpy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_sharded_latent_stats_world2(ragged):
    from oracle import oracle_np as onp
    n, D, world = 96, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, D, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((n, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((n, D)) - 1).astype(np.float32)
    z = (mu + rng.standard_normal((n, D)).astype(np.float32) * np.exp(0.5 * lv)).astype(np.float32)
    lq, pm = onp.approximate_log_q_z(mu, lv, z, "chen", 1000, True, False)
    lp = onp.sample_log_likelihood(z, reduce_batch_dim=False)
    lqx = onp.sample_log_likelihood(z, mu, lv, reduce_batch_dim=False)
    want = {"marginal_kl": (lq - lp).mean(), "mi": (lqx - lq).mean(), "tc": (lq - pm).mean(),
            "dim_kl": (pm - lp).mean(), "n_total": n, "m_total": n}
    lq_all = onp.approximate_log_q_z(mu, lv, z, "all", None, False, False)[0]
    for rank, res, comb, ok_gather, lq_local, (lo, hi) in results:
        assert ok_gather
        for k, v in want.items():
            assert abs(res[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, res[k], v)
        np.testing.assert_allclose(comb, lq_all, rtol=1e-5, atol=1e-4)     # float32 partials
        np.testing.assert_allclose(lq_local, lq[lo:hi], rtol=1e-12, atol=1e-9)
    assert results[0][1] == results[1][1]      # identical on every rank


def _oracle_kernel_sum(a, b, alphas, dist_eps, exclude_diag, diag_offset, symmetric=False):
    a64, b64 = a.numpy().astype(np.float64), b.numpy().astype(np.float64)
    d2 = dist_eps + np.abs((a64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None] - 2.0 * a64 @ b64.T)
    K = sum(np.exp(-al * d2) for al in alphas)
    i = np.arange(a64.shape[0])
    if exclude_diag:
        j = i + diag_offset
        ok = (j >= 0) & (j < b64.shape[0])
        K[i[ok], j[ok]] = 0.0
    if symmetric:      # tiles on (weight 1) and above (weight 2) the diagonal, as include/lvae.h describes
        assert diag_offset % 32 == 0
        cb = (diag_offset + i) // 32
        rb = np.arange(b64.shape[0]) // 32
        K = K * (2.0 * (rb[None, :] > cb[:, None]) + 1.0 * (rb[None, :] == cb[:, None]))
    return torch.tensor(K.sum(), dtype=torch.float64)


def _mmd_worker(rank, world, port, n1, n2, D, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvae import distributed as LD
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n1, D)).astype(np.float32)
    y = (0.8 * rng.standard_normal((n2, D)) + 0.1).astype(np.float32)
    xs = x[:n1 // 3] if rank == 0 else x[n1 // 3:]          # ragged shards
    ylo, yhi = LD.shard_bounds(n2, world, rank)
    alphas = [0.1 * i for i in range(15)]
    tts = LD.sharded_mmd(torch.from_numpy(xs.copy()), torch.from_numpy(y[ylo:yhi].copy()), alphas, 1e-5, True,
                         kernel_sum=_oracle_kernel_sum)
    biased = LD.sharded_mmd(torch.from_numpy(xs.copy()), torch.from_numpy(y[ylo:yhi].copy()), [1.0 / D ** 2], 0.0,
                            False, kernel_sum=_oracle_kernel_sum)
    q.put((rank, tts, biased))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mmd_world2():
    from oracle import oracle_np as onp
    n1, n2, D, world = 170, 153, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mmd_worker, args=(r, world, port, n1, n2, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n1, D)).astype(np.float32)
    y = (0.8 * rng.standard_normal((n2, D)) + 0.1).astype(np.float32)
    want_tts = onp.mmd_statistic_tts(x, y, [0.1 * i for i in range(15)])
    want_biased = onp.maximum_mean_discrepancy(y, x)
    for rank, tts, biased in results:
        assert abs(tts["mmd"] - want_tts) <= 1e-9 * max(1.0, abs(want_tts)), (tts, want_tts)
        assert abs(biased["mmd"] - want_biased) <= 1e-9 * max(1.0, abs(want_biased)), (biased, want_biased)
        assert tts["n1"] == n1 and tts["n2"] == n2
    assert results[0][1] == results[1][1]


def _oracle_permutation_sums(cols, col_offset, pooled, labels, alphas, dist_eps, symmetric):
    """This owner's share of (L^T K L, L^T K 1, 1^T K 1); symmetric: from the 32x32 tiles on and above the
    diagonal only, exactly as include/lvae.h describes for lvae_mmd_perm_f32."""
    c64, p64 = cols.numpy().astype(np.float64), pooled.numpy().astype(np.float64)
    nc, n = c64.shape[0], p64.shape[0]
    d2 = dist_eps + np.abs((c64 ** 2).sum(1)[:, None] + (p64 ** 2).sum(1)[None] - 2.0 * c64 @ p64.T)
    K = sum(np.exp(-al * d2) for al in alphas if al > 0)     # [nc, n]: K[c, r]
    i = np.arange(nc)
    K[i, i + col_offset] = 0.0
    L = (labels.numpy() != 0).astype(np.float64)            # [P, n]
    Lc = L[:, col_offset:col_offset + nc]                    # [P, nc]
    if not symmetric:
        q = np.einsum("pc,cr,pr->p", Lc, K, L)
        r = K.sum(1)
        return torch.from_numpy(q), torch.from_numpy(Lc @ r), torch.tensor(r.sum())
    assert col_offset % 32 == 0
    cb = (col_offset + i) // 32
    rb = np.arange(n) // 32
    strict = (rb[None, :] > cb[:, None]).astype(np.float64)
    diag = (rb[None, :] == cb[:, None]).astype(np.float64)
    Ks, Kd = K * strict, K * diag
    q = 2.0 * np.einsum("pc,cr,pr->p", Lc, Ks, L) + np.einsum("pc,cr,pr->p", Lc, Kd, L)
    lr = np.einsum("cr,pr->p", Ks + Kd, L) + Lc @ Ks.sum(1)
    tot = 2.0 * Ks.sum() + Kd.sum()
    return torch.from_numpy(q), torch.from_numpy(lr), torch.tensor(tot)


def _perm_worker(rank, world, port, n1, n2, D, P, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvae import distributed as LD
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n1, D)).astype(np.float32)
    y = (0.8 * rng.standard_normal((n2, D)) + 0.2).astype(np.float32)
    lab = np.zeros((P + 1, n1 + n2), dtype=np.uint8)
    lab[0, :n1] = 1
    for p in range(P):
        lab[1 + p, rng.permutation(n1 + n2)[:n1]] = 1
    xlo, xhi = LD.shard_bounds(n1, world, rank)
    ys = y[:n2 // 4] if rank == 0 else y[n2 // 4:]
    out = LD.sharded_mmd_permutation_test(torch.from_numpy(x[xlo:xhi].copy()), torch.from_numpy(ys.copy()),
                                          [0.1 * i for i in range(15)], torch.from_numpy(lab), 1e-5,
                                          permutation_sums=_oracle_permutation_sums)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mmd_permutation_world2():
    from oracle import oracle_np as onp
    n1, n2, D, P, world = 141, 136, 8, 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_perm_worker, args=(r, world, port, n1, n2, D, P, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n1, D)).astype(np.float32)
    y = (0.8 * rng.standard_normal((n2, D)) + 0.2).astype(np.float32)
    lab = np.zeros((P + 1, n1 + n2), dtype=np.uint8)
    lab[0, :n1] = 1
    for p in range(P):
        lab[1 + p, rng.permutation(n1 + n2)[:n1]] = 1
    alphas = [0.1 * i for i in range(15)]
    obs, K = onp.mmd_statistic_tts(x, y, alphas, ret_matrix=True)
    ref = onp.mmd_permutation_stats(K, n1, n2, 1 - lab)
    for rank, out in results:
        assert abs(out["statistic"] - obs) <= 1e-9 * max(1.0, abs(obs))
        np.testing.assert_allclose(out["statistics"], ref[1:], rtol=1e-8, atol=1e-11)
        assert abs(out["pval"] - float((ref[1:] >= ref[0]).mean())) < 1e-12
    assert results[0][1] == results[1][1]


class _OracleBlockKernels:
    """CPU stand-in for distributed.HipBlockKernels (NumPy float64): partial logsumexp over a subset of
    posteriors, the merge of two partials, and the five float64 sums."""

    @staticmethod
    def partial(mu, logvar, z):
        from oracle import oracle_np as onp
        dens = onp.log_prob_pairs(z.numpy(), mu.numpy(), np.exp(logvar.numpy().astype(np.float64)))   # [cols, rows]
        m = dens.max(0)
        return torch.from_numpy(m), torch.from_numpy(np.exp(dens - m[None]).sum(0))

    @staticmethod
    def combine(own, rest, log_norm):
        m, s = own[0].numpy(), own[1].numpy()
        if rest is not None:
            m2, s2 = rest[0].numpy(), rest[1].numpy()
            mm = np.maximum(m, m2)
            s = s * np.exp(m - mm) + s2 * np.exp(m2 - mm)
            m = mm
        if log_norm is None:             # merged (max, sum) pair, like estimators.lse_combine
            return torch.from_numpy(m), torch.from_numpy(s)
        return torch.from_numpy(m + np.log(s) - log_norm)

    @staticmethod
    def sums(z_local, mu_local, logvar_local, log_q_z):
        from oracle import oracle_np as onp
        z = z_local.numpy()
        lp = onp.sample_log_likelihood(z, reduce_batch_dim=False)
        lqx = onp.sample_log_likelihood(z, mu_local.numpy(), logvar_local.numpy(), reduce_batch_dim=False)
        return torch.tensor([float(log_q_z.sum()), 0.0, lp.sum(), lqx.sum(), float(len(z))], dtype=torch.float64)


def _overlap_worker(rank, world, port, n, D, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvae import distributed as LD
    rng = np.random.default_rng(1)
    mu = rng.standard_normal((n, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((n, D)) - 1).astype(np.float32)
    z = (mu + rng.standard_normal((n, D)).astype(np.float32) * np.exp(0.5 * lv)).astype(np.float32)
    lo, hi = LD.shard_bounds(n, world, rank)
    t = lambda a: torch.from_numpy(a[lo:hi].copy())
    counts = [n // world] * world
    plain = LD.sharded_latent_stats(t(mu), t(lv), t(z), method="all", local_stats=_oracle_local_stats, counts=counts)
    over = LD.sharded_latent_stats(t(mu), t(lv), t(z), method="all", counts=counts, overlap=True,
                                   block_kernels=_OracleBlockKernels)
    auto = LD.sharded_latent_stats(t(mu), t(lv), t(z), method="chen", dataset_size=500, counts=counts,
                                   block_kernels=_OracleBlockKernels)                 # overlap chosen by default
    # the device-side half on its own, three calls queued before anything is read (bench.py's step): the
    # all-reduce of each call is asynchronous and only the last one's handle is waited for
    pending = [LD.sharded_latent_sums(t(mu), t(lv), t(z), method="all", counts=counts, overlap=True,
                                      block_kernels=_OracleBlockKernels, async_reduce=True) for _ in range(3)]
    sums, lq, n_tot, work = pending[-1]
    if work is not None:
        work.wait()
    queued = LD.stats_from_sums(sums, n_tot, lq)
    err = None
    try:
        LD.sharded_latent_stats(t(mu), t(lv), t(z), prod_marginals=True, counts=counts, overlap=True,
                                block_kernels=_OracleBlockKernels)
    except ValueError as e:
        err = str(e)
    if rank == 0:
        q.put({"plain": {k: v for k, v in plain.items() if isinstance(v, float)},
               "over": {k: v for k, v in over.items() if isinstance(v, float)},
               "queued": {k: v for k, v in queued.items() if isinstance(v, float)},
               "auto_marginal_kl": auto["marginal_kl"], "lq_plain": plain["log_q_z_local"].numpy(),
               "lq_over": over["log_q_z_local"].numpy(), "err": err})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_exchange_matches_plain(world):
    """The all-gather hidden behind the local block (own posteriors first, the others after the wait, partial
    logsumexps merged) gives the same statistics as gather-then-compute, for every rank position."""
    n, D = 96 * world, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, n, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, v in res["plain"].items():
        assert abs(res["over"][k] - v) <= 1e-9 * max(1.0, abs(v)), k
        assert res["queued"][k] == res["over"][k], k
    np.testing.assert_allclose(res["lq_over"], res["lq_plain"], rtol=1e-10, atol=1e-10)
    assert abs(res["auto_marginal_kl"] - (res["plain"]["marginal_kl"] - math.log(500))) < 1e-9
    assert res["err"] and "overlap=True needs" in res["err"]


def _lse_far_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvae import distributed as LD
    rng = np.random.default_rng(40)
    M = 257
    # every rank's partials for ALL rows; the maxima of the ranks lie up to ~200 nats apart, row by row in a
    # different order, so that every rank is the dominant one somewhere and flushed to zero somewhere else
    maxes = rng.standard_normal((world, M)) * 5.0 + np.array([0.0, -60.0, -200.0])[:world, None] * rng.choice([1.0, -1.0], (1, M))
    sums = rng.uniform(1.0, 50.0, (world, M))
    out = LD.combine_partial_lse(torch.from_numpy(maxes[rank]).float(), torch.from_numpy(sums[rank]).float(),
                                 math.log(1000.0), kernels=_OracleLseKernels)
    if rank == 0:
        q.put((out.numpy(), maxes.astype(np.float32).astype(np.float64), sums.astype(np.float32).astype(np.float64)))
    dist.barrier()
    dist.destroy_process_group()


def test_combine_partial_lse_with_rank_maxima_far_apart():
    """The column-sharded merge exchanges float32 sums re-based on the global maximum (lvae_lse_rebase_f32): a rank
    whose maximum lies > 87 nats below the global one contributes exactly 0, the others round at 2^-24 per add.
    Against the float64 logsumexp of the same float32 partials: 1e-6."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lse_far_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, maxes, sums = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gm = maxes.max(0)
    want = gm + np.log((sums * np.exp(maxes - gm[None])).sum(0)) - math.log(1000.0)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
