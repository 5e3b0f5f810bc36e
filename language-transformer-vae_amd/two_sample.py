"""Mirror of the one torch-two-sample entry the reference uses:
``torch_two_sample.statistics_diff.MMDStatistic`` (call site NewsVAE/loss_and_optimisation.py:617-619).

PARITY UNPINNED: josipd/torch-two-sample is an un-vendored submodule with no pinned revision
(.gitmodules:4-6); this follows its published algorithm -- pooled Euclidean distances
sqrt(1e-5 + |a.a + b.b - 2 a.b|), kernel sum_alpha exp(-alpha d^2), unbiased statistic with the
diagonals removed.  ``pval`` (never called by the reference) runs the permutation test from the samples
of the last call through the fused kernels; the only ``distances`` argument it accepts is the matrix that call
returned (``ret_matrix=True``) -- any other matrix raises, since it would not be the one the result comes from.
"""
import torch

from . import _native as N
from . import estimators as E

PDIST_EPS = 1e-5


class MMDStatistic:
    def __init__(self, n_1, n_2):
        self.n_1 = n_1
        self.n_2 = n_2
        self.a00 = 1. / (n_1 * (n_1 - 1))
        self.a11 = 1. / (n_2 * (n_2 - 1))
        self.a01 = - 1. / (n_1 * n_2)

    def __call__(self, sample_1, sample_2, alphas, ret_matrix=False):
        if sample_1.shape[0] != self.n_1 or sample_2.shape[0] != self.n_2:
            raise ValueError("sample sizes differ from the ones the statistic was built for")
        self._last = (sample_1, sample_2, list(alphas))
        self._last_matrix = None
        out = E.mmd_rbf(sample_1, sample_2, list(alphas), PDIST_EPS, unbiased=True)
        mmd = out[0].float()
        if ret_matrix:
            s12 = torch.cat((N.as_f32(sample_1, "sample_1"), N.as_f32(sample_2, "sample_2")), 0)
            self._last_matrix = E._kernel_matrix(s12, s12, list(alphas), PDIST_EPS)
            return mmd, self._last_matrix
        return mmd

    def pval(self, distances=None, n_permutations=1000, generator=None):
        """Permutation-test p-value for the samples of the last ``__call__`` (torch-two-sample passes the
        kernel matrix returned with ``ret_matrix=True``; here it is only a token -- the n x n matrix is
        never read, the label sums are regenerated from the samples on the matrix cores)."""
        if getattr(self, "_last", None) is None:
            raise RuntimeError("call the statistic on two samples first")
        if distances is not None:
            mine = getattr(self, "_last_matrix", None)
            same = mine is not None and (distances is mine or (
                isinstance(distances, torch.Tensor) and distances.shape == mine.shape
                and distances.data_ptr() == mine.data_ptr()))
            if not same:
                raise ValueError("pval(distances): only the kernel matrix returned by the last call with "
                                 "ret_matrix=True is accepted (the label sums are regenerated from that call's "
                                 "samples; another matrix would silently be ignored)")
        s1, s2, alphas = self._last
        return float(E.mmd_permutation_test(s1, s2, alphas, n_permutations, PDIST_EPS,
                                            generator=generator)["pval"])
