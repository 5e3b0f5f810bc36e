import math, os, sys
sys.path.insert(0, "/root/repo")
import torch
from lvae import estimators as E
n = int(sys.argv[1]); D = 32
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(n, D, device="cuda", generator=g)
lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
for it in range(6):
    E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    torch.cuda.synchronize()
print("done")
