"""MMD permutation test timing (BASELINE configs[3]: N = M = 262,144, D = 32, 100 permutations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

PATHS = {"auto": 0, "fp32": 1, "bf16x3": 3}

def run(n1, n2, D, P, iters=2, path=0):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n1, D, device="cuda", generator=g)
    y = 0.98 * torch.randn(n2, D, device="cuda", generator=g) + 0.01
    alphas = [0.1 * i for i in range(15)]
    labels = E.permutation_labels(n1, n2, P, "cuda", g)
    E.mmd_permutation_test(x[:4096], y[:4096], alphas, labels=E.permutation_labels(4096, 4096, 4, "cuda", g))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = E.mmd_permutation_test(x, y, alphas, labels=labels, path=path)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        plain = E.mmd_rbf(x, y, alphas, 1e-5, unbiased=True, path=path)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t0) / iters
    n = n1 + n2
    print(f"[path {path}] MMD perm n1={n1} n2={n2} D={D} P={P}+1: {dt*1e3:.1f} ms  {n*n/dt:.3e} pooled entries/s "
          f"({0.5*n*n*2*D/dt/157.3e12*100:.1f}% of fp32 MFMA peak on the distance part: upper triangle, 2D flop per entry); statistic alone "
          f"{dp*1e3:.1f} ms; stat={float(out['statistic']):.3e} (plain {float(plain[0]):.3e}) pval={float(out['pval']):.3f}",
          flush=True)

if __name__ == "__main__":
    # LVAE_PAIR_MATH = auto (default: what LVAE_PATH_AUTO picks) | fp32 | bf16x3 | all
    mode = os.environ.get("LVAE_PAIR_MATH", "auto")
    for a in sys.argv[1:]:
        for name in (PATHS if mode == "all" else (mode,)):
            run(*(int(v) for v in a.split(",")), path=PATHS[name])
