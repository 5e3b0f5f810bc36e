"""Randomised shape sweep of the fused training-time tail (lvae_encode_tail_ws_f32 / _bwd_ws_f32: one workgroup and
B / 4 workgroups) against the same tail on the general kernels, forward values and the gradients of a random linear
functional of all outputs.  Run by name on the GPU box -- `python tests/fuzz_tail.py <cases> <seed>` -- and not
collected by pytest; prints every failing case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lvae import encoder_ops as EO

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
bad, worst = 0, 0.0
KEYS = ("latent_z", "log_q_z_x", "log_p_z", "log_q_z", "log_q_z_prod_marg", "mmd_loss")


def run(form, mu0, lv0, eps, prior, coef):
    mu, lv = mu0.clone().requires_grad_(True), lv0.clone().requires_grad_(True)
    out = EO.encode_tail(mu, lv, n_samples=eps.shape[1], dataset_size=4321, eps=eps, prior_sample=prior, fused=form)
    loss = sum((coef[k] * out[k]).sum() for k in KEYS)
    loss.backward()
    return {k: out[k].detach().double().cpu() for k in KEYS}, mu.grad.double().cpu(), lv.grad.double().cpu()


for c in range(cases):
    D = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64]))
    K = int(rng.choice([1, 1, 1, 2, 3, 5]))
    B = int(rng.choice([2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 500, 777, 1000, 1024, int(rng.integers(2, 1100))]))
    g = torch.Generator(device="cuda").manual_seed(seed * 100003 + c)
    mu0 = torch.randn(B, D, device="cuda", generator=g) * float(rng.choice([0.3, 1.0, 2.0]))
    lv0 = float(rng.choice([0.2, 0.5, 1.0])) * torch.randn(B, D, device="cuda", generator=g) + float(rng.choice([-2.0, -1.0, 0.0]))
    eps = torch.randn(B, K, D, device="cuda", generator=g)
    prior = torch.randn(B, D, device="cuda", generator=g)
    coef = {k: torch.randn((), device="cuda", generator=g) for k in KEYS}
    coef["latent_z"] = torch.randn(B, K, D, device="cuda", generator=g) * 0.1
    coef["log_q_z_x"] = torch.randn(B, K, device="cuda", generator=g) * 0.1
    coef["log_p_z"] = torch.randn(B, K, device="cuda", generator=g) * 0.1
    ref = run(False, mu0, lv0, eps, prior, coef)
    forms = ["multi_wg"] + (["one_wg"] if EO.tiny_tail_supported(B, K, D) else [])
    for form in forms:
        if not EO.fused_tail_supported(B, K, D):
            continue
        got = run(form, mu0, lv0, eps, prior, coef)
        tag = f"case {c}: form={form} B={B} K={K} D={D}"
        for k in KEYS:
            e = float((got[0][k] - ref[0][k]).abs().max() / max(1.0, float(ref[0][k].abs().max())))
            worst = max(worst, e)
            if not np.isfinite(e) or e > 2e-5:
                print(f"FAIL forward {k} rel err {e:.2e}  {tag}", flush=True)
                bad += 1
        for name, a, b in (("d_mu", got[1], ref[1]), ("d_logvar", got[2], ref[2])):
            e = float((a - b).abs().max() / max(1e-6, float(b.abs().max())))
            worst = max(worst, e)
            if not np.isfinite(e) or e > 1e-4:
                print(f"FAIL {name} rel err {e:.2e}  {tag}", flush=True)
                bad += 1
    if (c + 1) % 25 == 0:
        print(f"{c + 1} cases, {bad} failures, worst rel err {worst:.2e}", flush=True)
print(f"{cases} cases, {bad} failures, worst rel err {worst:.2e}", flush=True)
sys.exit(1 if bad else 0)
