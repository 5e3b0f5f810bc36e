"""Error of the three joint-density paths (fp32 MFMA, direct VALU, split-bf16 MFMA) against an fp64 evaluation
of the same formula (torch, on the GPU, chunked), on one seeded input.  Prints max / rms absolute error of
log q(z_i) and the relative error of the mean."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

def fp64_log_qz(mu, lv, z, chunk=128):
    mu, lv, z = mu.double(), lv.double(), z.double()
    out = []
    c = -0.5 * (math.log(2 * math.pi) + lv)                       # [N, D]
    p = torch.exp(-lv)
    for i in range(0, z.shape[0], chunk):
        d = z[i:i + chunk, None, :] - mu[None]
        out.append(torch.logsumexp((c[None] - 0.5 * d * d * p[None]).sum(-1), dim=1))
    return torch.cat(out)

def fp64_log_qz_pm(mu, lv, z, chunk=64):
    mu, lv, z = mu.double(), lv.double(), z.double()
    out = []
    c = -0.5 * (math.log(2 * math.pi) + lv)
    p = torch.exp(-lv)
    for i in range(0, z.shape[0], chunk):
        d = z[i:i + chunk, None, :] - mu[None]
        out.append(torch.logsumexp(c[None] - 0.5 * d * d * p[None], dim=1).sum(-1))
    return torch.cat(out)

for n, D, off in ((8192, 32, 0.0), (8192, 16, 0.0), (8192, 32, 3.0), (4096, 20, 0.0)):
    g = torch.Generator(device="cuda").manual_seed(2)
    mu = torch.randn(n, D, device="cuda", generator=g) + off
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    ref = fp64_log_qz(mu, lv, z)
    for name, path in (("fp32 MFMA", 1), ("direct", 2), ("bf16x3", 3)):
        lq, _ = E.log_qz_raw(mu, lv, z, 0.0, False, path=path)
        e = lq.double() - ref
        print(f"n={n} D={D} offset={off}: {name:10s} max|err| {float(e.abs().max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}"
              f"  rel err of mean {abs(float(e.mean()) / float(ref.mean())):.2e}", flush=True)
    # prod-marginals: AUTO (bf16 matrix-pipe polynomial from 2^22 pairs) against the direct packed-fp32 sweep
    refpm = fp64_log_qz_pm(mu, lv, z)
    for name, path in (("auto (matrix pipe)", 0), ("direct", 2)):
        _, pm = E.log_qz_raw(mu, lv, z, 0.0, True, path=path)
        e = pm.double() - refpm
        print(f"n={n} D={D} offset={off}: prod-marginals {name:18s} max|err| {float(e.abs().max()):.3e}  rms "
              f"{float(e.pow(2).mean().sqrt()):.3e}  rel err of mean {abs(float(e.mean()) / float(refpm.mean())):.2e}", flush=True)
