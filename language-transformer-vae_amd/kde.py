"""Mirror of NewsVAE/utils_latent_analysis_optimisation.py (GaussianKDE) over the HIP KDE kernel."""
import torch

from . import _native as N


class GaussianKDE:
    """utils_latent_analysis_optimisation.py:11-59.  X [n, d] fit points; bw numeric or "scott"
    (n ** (-1 / (d + 4)), :24-26).  ``device`` is accepted for signature parity; tensors stay where X is."""

    def __init__(self, X, bw, device=None):
        self.X = X
        if bw == "scott":
            n, d = X.shape
            bw = n ** (-1. / (d + 4))
        self.bw = bw
        self.dims = X.shape[-1]
        self.n = X.shape[0]

    def score_samples(self, Y, X=None, stabilise=False):
        """:33-60 -- log kernel density estimate of every row of Y [m, d]; unstabilised like the
        reference (quirk Q6: -inf when every kernel term underflows) unless ``stabilise`` is set.  A batch ``X``
        is normalised by ``self.n``, the size of the fit set, like the reference (:58)."""
        if X is None:
            X = self.X
        Xc, Yc = N.as_f32(X, "X"), N.as_f32(Y, "Y")
        if Xc.dim() != 2 or Yc.dim() != 2 or Xc.shape[1] != Yc.shape[1]:
            raise ValueError("X and Y must be [n, d] and [m, d]")
        dev = Yc.device
        with torch.cuda.device(dev):
            out = torch.empty(Yc.shape[0], dtype=torch.float32, device=dev)
            nbytes = N._i64(0)
            N.check(N.lib().lvae_kde_workspace_bytes(Xc.shape[0], Yc.shape[0], nbytes))
            ws = N.workspace(nbytes.value, dev)
            N.check(N.lib().lvae_kde_score_f32(N.ptr(Xc), N.ptr(Yc), Xc.shape[0], int(self.n), Yc.shape[0], Xc.shape[1],
                                               float(self.bw), int(bool(stabilise)), N.ptr(out), N.ptr(ws),
                                               nbytes.value, N.stream_ptr(dev)))
        return out
