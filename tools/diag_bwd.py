"""Diagnostic: error of the matrix-core backward against float64 as the posteriors sharpen, next to the gate's
conditioning estimate (pair_grad_mfma.hip: largest eight-dimension group of 0.5 (2|mu'|/sigma + 3)^2, times
sqrt(groups)); and small-problem timings of the two paths."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_grad_mfma import _problem, _reference_grads, _our_grads


def estimate(mu, lv):
    D = mu.shape[1]
    DP = 16 if D <= 16 else 32 if D <= 32 else 64 if D <= 64 else 128
    m = mu - mu.mean(0, keepdim=True)
    t = 0.5 * (2 * m.abs() * torch.exp(-0.5 * lv) + 3) ** 2
    t = torch.nn.functional.pad(t, (0, DP - D))
    return float(t.reshape(mu.shape[0], DP // 8, 8).sum(2).max() * math.sqrt(DP / 8))


for D in (32, 128):
    for sc in (1.0, 0.5, 0.25, 0.1, 0.05, 0.02):
        mu, lv, z, a = _problem(D, 1500, 1400, D, sigma_scale=sc)
        want = _reference_grads(mu, lv, z, a, math.log(1400))
        row = [f"D={D} sigma x{sc}: estimate {estimate(mu, lv):.3g}"]
        for path in ("mfma", "valu"):
            os.environ["LVAE_BWD_PATH"] = path
            __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
            got = _our_grads(mu, lv, z, a)
            row.append(path + " " + " ".join(f"{float((g.double() - w).abs().max() / w.abs().max()):.1e}" for g, w in zip(got, want)))
        print("; ".join(row), flush=True)

from bench_bwd import bwd  # noqa: E402
from lvae import estimators as E  # noqa: E402
for n in (160, 256, 512, 1024, 2048):
    mu, lv, z, a = _problem(1, n, n, 32)
    lq, _ = E.log_qz_raw(mu, lv, z, math.log(n), False)
    out = []
    for path in ("mfma", "valu", "auto"):
        os.environ["LVAE_BWD_PATH"] = path
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
        bwd(z, mu, lv, math.log(n), lq, a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            bwd(z, mu, lv, math.log(n), lq, a)
        torch.cuda.synchronize()
        out.append(f"{path} {(time.perf_counter() - t0) / 20 * 1e6:.0f} us")
    print(f"n={n} D=32:", ", ".join(out), flush=True)
