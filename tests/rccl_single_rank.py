"""One rank of a ONE-rank RCCL job (backend "nccl" on a single MI355X): the collective calls of
lvae.distributed -- asynchronous all-gather with `work.wait()`, asynchronous all-reduce whose handle is held for
two steps, padded all-gather, max / sum all-reduces of the column-sharded form -- issued through the real
ProcessGroupNCCL (its own stream, its tensor stash) instead of gloo.  With one rank every collective is a copy, so
the results must equal the single-GPU calls.  Started by tests/test_gpu_rccl.py as a fresh process; prints OK.

(Several ranks cannot share one GPU under RCCL, so this is as far as a one-GPU box goes; the multi-rank logic is
covered by the gloo tests.)"""
import collections
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from lvae import distributed as LD
    from lvae import estimators as E

    n, D = 4096, 32
    g = torch.Generator(device=dev).manual_seed(5)
    mu = torch.randn(n, D, device=dev, generator=g)
    logvar = 0.5 * torch.randn(n, D, device=dev, generator=g) - 1.0
    z = mu + torch.randn(n, D, device=dev, generator=g) * torch.exp(0.5 * logvar)
    log_norm = math.log(n)
    want_sums, want_lq = LD.hip_local_stats(mu, logvar, z, mu, logvar, log_norm, False)
    want_sums, want_lq = want_sums.cpu(), want_lq.cpu()

    # 1. the overlapped exchange: async all-gather on RCCL's stream, own block meanwhile, stream-side wait
    #    (mu and logvar are gathered as two collectives and read in place)
    tl = LD.Timeline(dev)
    for it in range(3):
        sums, lq, n_total = LD.overlapped_stats(mu, logvar, z, None, lambda nt: math.log(nt),
                                                timeline=tl if it == 2 else None)
    phases = tl.phases_ms()
    assert "own_block" in phases and "wait_all_gather" in phases, phases
    assert n_total == n
    assert torch.allclose(lq.cpu(), want_lq, rtol=0, atol=2e-5), float((lq.cpu() - want_lq).abs().max())
    assert torch.allclose(sums.cpu(), want_sums, rtol=1e-6), (sums.cpu(), want_sums)   # (two-block logsumexp)

    # 2. the bench's step pattern: async all-reduce, handles retired two steps late, no host wait in between
    pending = collections.deque()
    for step in range(8):
        cur = LD.sharded_latent_sums(mu, logvar, z, method="all", counts=[n], async_reduce=True)
        assert len(cur) == 4
        pending.append(cur)
        while len(pending) > 2:
            e = pending.popleft()
            if e[3] is not None:
                e[3].wait()
    last = None
    while pending:
        last = pending.popleft()
        if last[3] is not None:
            last[3].wait()
    torch.cuda.synchronize()
    assert torch.allclose(last[0].cpu(), want_sums, rtol=1e-9)
    out = LD.stats_from_sums(last[0], last[2], last[1])
    assert abs(out["marginal_kl"] - float((want_sums[0] - want_sums[2]) / want_sums[4])) < 1e-9

    # 3. blocking form with the size exchange (counts unknown) and prod-marginals
    full = LD.sharded_latent_stats(mu[:512], logvar[:512], z[:512], method="chen", dataset_size=42068,
                                   prod_marginals=True)
    lq1, pm1 = E.approximate_log_q_z(mu[:512], logvar[:512], z[:512], method="chen", dataset_size=42068,
                                     prod_marginals=True, reduce_mean=True)
    assert abs(full["log_q_z"] - float(lq1)) < 1e-4 * abs(float(lq1))
    assert abs(full["log_q_z_prod_marg"] - float(pm1)) < 1e-4 * abs(float(pm1))

    # 4. column-sharded form: all-reduce(max), rebase, all-reduce(sum), finish
    rmax, rsum = E.log_qz_partial(mu, logvar, z)
    lq2 = LD.combine_partial_lse(rmax, rsum, log_norm)
    assert torch.allclose(lq2.cpu(), want_lq, rtol=0, atol=2e-5)

    # 5. MMD and IW shards
    x, y = z[:1024].contiguous(), torch.randn(1024, D, device=dev, generator=g)
    alphas = [0.1 * i for i in range(15)]
    got = LD.sharded_mmd(x, y, alphas, dist_eps=1e-5, unbiased=True)
    ref = float(E.mmd_rbf(x, y, alphas, 1e-5, True)[0])
    assert abs(got["mmd"] - ref) <= 1e-4 * max(abs(ref), 1e-3), (got["mmd"], ref)
    B, K = 64, 50
    eps = torch.randn(B, K, D, device=dev, generator=g)
    lpxz = 5 * torch.randn(B, K, device=dev, generator=g) - 100
    iw = LD.sharded_iw_log_p_x(mu[:B], logvar[:B], eps, lpxz)
    ref_iw = E.reparam_iw(mu[:B], logvar[:B], eps, lpxz, return_z=False)["iw_ll"]
    assert torch.equal(iw.cpu(), ref_iw.cpu())

    dist.barrier()
    dist.destroy_process_group()
    print("OK", flush=True)


if __name__ == "__main__":
    main()
