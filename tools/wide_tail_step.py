"""One training-time tail (encode_tail + KL, forward + backward) at a wide latent, for kernel traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import encoder_ops as EO, estimators as E
B, D = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(1)
mu = torch.randn(B, D, device="cuda", generator=g).requires_grad_(True)
lv = (0.5 * torch.randn(B, D, device="cuda", generator=g) - 1.0).requires_grad_(True)
eps = torch.randn(B, 1, D, device="cuda", generator=g)
for _ in range(5):
    mu.grad = lv.grad = None
    o = EO.encode_tail(mu, lv, n_samples=1, eps=eps)
    kl, _ = E.kl_divergence(mu, lv, 0.5)
    (o["log_q_z_x"].mean() - o["log_q_z"] + o["log_q_z_prod_marg"] - o["log_p_z"].mean() + o["mmd_loss"] + kl).backward()
torch.cuda.synchronize()
