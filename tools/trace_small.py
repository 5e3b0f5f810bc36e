"""A few calls of the small-problem paths for a rocprofv3 kernel trace (gaps between launches):
trace_small.py {stats|tail} n"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import distributed as LD, encoder_ops as EO, estimators as E
mode, n = sys.argv[1], int(sys.argv[2])
D = 32
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(n, D, device="cuda", generator=g)
lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
eps = torch.randn(n, 1, D, device="cuda", generator=g)
z = mu + eps[:, 0] * torch.exp(0.5 * lv)
prior = torch.randn(n, D, device="cuda", generator=g)
for it in range(6):
    if mode == "stats":
        LD.hip_local_stats(mu, lv, z, mu, lv, math.log(n), False)
    else:
        EO.encode_tail(mu, lv, 1, 10000, eps=eps, prior_sample=prior)
        E.kl_divergence(mu, lv, 0.5)
    torch.cuda.synchronize()
print("done", flush=True)
