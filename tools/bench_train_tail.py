"""Latency of the training-time tail (encode_tail + KL, forward and backward) at the reference's batch
sizes, next to the same op sequence in PyTorch eager on the same GPU (oracle/torch_eager-style formulas)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import encoder_ops as EO, estimators as E

LOG2PI = math.log(2 * math.pi)

def eager_tail(mu, logvar, eps, dataset_size=10000):
    std = logvar.mul(0.5).exp()
    z = mu.unsqueeze(1) + eps * std.unsqueeze(1)
    single = z[:, 0, :]
    lqx = (-0.5 * (logvar.unsqueeze(1) + LOG2PI + (z - mu.unsqueeze(1)) ** 2 / logvar.exp().unsqueeze(1))).sum(-1)
    lpz = (-0.5 * (LOG2PI + z ** 2)).sum(-1)
    B, D = mu.shape
    dens = -0.5 * (logvar.unsqueeze(0) + LOG2PI + (single.unsqueeze(1) - mu.unsqueeze(0)) ** 2 / logvar.exp().unsqueeze(0))
    norm = math.log(B * dataset_size)
    lq = (torch.logsumexp(dens.sum(2), 1) - norm).mean()
    lqpm = (torch.logsumexp(dens, 1) - norm).sum(1).mean()
    kl = (0.5 * (mu ** 2 + logvar.exp() - logvar - 1)).sum(1).mean()
    return lqx.mean() - lq + 3.0 * (lq - lqpm) + 0.5 * (lqpm - lpz.mean()) + 0.25 * kl

FUSED = [None]

def hip_tail(mu, logvar, eps):
    out = EO.encode_tail(mu, logvar, n_samples=eps.shape[1], dataset_size=10000, eps=eps, fused=FUSED[0])
    kl, _ = E.kl_divergence(mu, logvar, 0.5)
    return (out["log_q_z_x"].mean() - out["log_q_z"] + 3.0 * (out["log_q_z"] - out["log_q_z_prod_marg"])
            + 0.5 * (out["log_q_z_prod_marg"] - out["log_p_z"].mean()) + 0.25 * kl)

def timeit(fn, mu, logvar, eps, iters=50):
    for _ in range(5):
        mu.grad = logvar.grad = None
        fn(mu, logvar, eps).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        mu.grad = logvar.grad = None
        fn(mu, logvar, eps).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

def fwd_only(mu, logvar, eps, fused):
    with torch.no_grad():
        f = lambda: EO.encode_tail(mu, logvar, n_samples=1, dataset_size=10000, eps=eps, fused=fused)
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 50 * 1e6

for B in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    g = torch.Generator(device="cuda").manual_seed(B)
    mu = torch.randn(B, 32, device="cuda", generator=g).requires_grad_(True)
    logvar = (0.5 * torch.randn(B, 32, device="cuda", generator=g) - 1).requires_grad_(True)
    eps = torch.randn(B, 1, 32, device="cuda", generator=g)
    line = f"B={B}:"
    ref_grad = None
    for name, fused in (("default", None), ("one_wg", "one_wg"), ("multi_wg", "multi_wg"), ("general", False)):
        if fused == "one_wg" and not EO.tiny_tail_supported(B, 1, 32):
            continue
        FUSED[0] = fused
        t = timeit(hip_tail, mu, logvar, eps)
        if ref_grad is None:
            ref_grad = mu.grad.clone()
        err = float((ref_grad - mu.grad).abs().max() / mu.grad.abs().max())
        line += f"  {name} fwd+bwd {t:.0f} us, fwd only {fwd_only(mu, logvar, eps, fused):.0f} us (grad diff {err:.0e})"
    b = timeit(eager_tail, mu, logvar, eps)
    print(line + f"   PyTorch eager (GPU) {b:.0f} us", flush=True)
