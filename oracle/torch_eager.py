"""PyTorch-eager CPU restatement of the reference's op chain for log q(z) -- the "reference CPU path"
that bench.py times beside the GPU numbers.  TEST/BENCH INFRASTRUCTURE ONLY.

Same tensor algebra as approximate_log_q_z (NewsVAE/loss_and_optimisation.py:178-215): broadcast
[rows,1,D] against [1,N,D], materialise the log-density tensor, sum / logsumexp.  The reference
materialises all M rows at once; here rows are chunked so that N = 65,536 fits in memory, which is
exact because both logsumexps run over the posterior axis only.  Pinned to the golden vectors in
tests/test_oracle_golden.py.
"""
import math

import torch

LOG_2PI = math.log(2 * math.pi)


def log_qz_rows(mu, logvar, z_rows, log_norm, prod_marginals=False):
    var = logvar.exp().unsqueeze(0)                                               # :95
    log_dens = -0.5 * (logvar.unsqueeze(0) + LOG_2PI + (z_rows.unsqueeze(1) - mu.unsqueeze(0)) ** 2 / var)  # :96
    log_q_z = torch.logsumexp(log_dens.sum(dim=2), dim=1) - log_norm              # :198
    pm = (torch.logsumexp(log_dens, dim=1) - log_norm).sum(dim=1) if prod_marginals else None   # :195
    return log_q_z, pm


def log_qz(mu, logvar, z, log_norm, prod_marginals=False, row_chunk=64):
    outs, pms = [], []
    for r0 in range(0, z.shape[0], row_chunk):
        a, b = log_qz_rows(mu, logvar, z[r0:r0 + row_chunk], log_norm, prod_marginals)
        outs.append(a)
        pms.append(b)
    return torch.cat(outs), (torch.cat(pms) if prod_marginals else None)
