import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E
B, D = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(1)
mu = torch.randn(B, D, device="cuda", generator=g).requires_grad_(True)
lv = (0.5 * torch.randn(B, D, device="cuda", generator=g) - 1.0).requires_grad_(True)
z = (mu + torch.randn(B, D, device="cuda", generator=g) * torch.exp(0.5 * lv)).detach()
for _ in range(5):
    mu.grad = lv.grad = None
    lq, pm = E.approximate_log_q_z(mu, lv, z, method="chen", dataset_size=42068, prod_marginals=True, reduce_mean=True)
    (lq + 0.5 * pm).backward()
torch.cuda.synchronize()
