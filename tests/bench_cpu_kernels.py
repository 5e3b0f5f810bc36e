"""Test hook for bench.py (LVAE_BENCH_KERNELS=tests.bench_cpu_kernels): oracle-backed per-rank compute so that the
CPU suite can run the benchmark's launch / sharding / exchange / reporting code over gloo without a GPU.
Test infrastructure only -- the numbers such a run prints mean nothing."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from test_distributed_gloo import _OracleBlockKernels, _oracle_local_stats, _oracle_permutation_sums  # noqa: E402


def _maybe_fail():
    if os.environ.get("LVAE_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise RuntimeError("injected failure of the exchange step (test)")


def local_stats(*a, **k):
    _maybe_fail()
    return _oracle_local_stats(*a, **k)


class BlockKernels(_OracleBlockKernels):
    @staticmethod
    def partial(mu, logvar, z):
        _maybe_fail()
        return _OracleBlockKernels.partial(mu, logvar, z)


def permutation_sums(cols, col_offset, pooled, labels, alphas, dist_eps, symmetric):
    _maybe_fail()
    return _oracle_permutation_sums(cols, col_offset, pooled, labels, alphas, dist_eps, symmetric)


def reparam_iw(mu, logvar, eps, log_p_x_z):
    _maybe_fail()
    from oracle import oracle_np as onp
    z = onp.reparameterize(mu.numpy(), logvar.numpy(), eps.numpy())
    lq = onp.sample_log_likelihood(z, mu.numpy(), logvar.numpy(), reduce_batch_dim=False)
    lp = onp.sample_log_likelihood(z, reduce_batch_dim=False)
    return torch.from_numpy(onp.iw_log_p_x(log_p_x_z.numpy(), lp, lq).astype("float32"))
