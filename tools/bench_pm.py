"""Prod-marginals sweep at a given size: kernel-level timing through lvae_profile_begin/end is for the
joint kernels, so this times whole calls and (optionally, under rocprofv3) the kernels by name."""
import sys, time, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

def run(N, M, D, iters=3):
    g = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.randn(N, D, device="cuda", generator=g)
    logvar = 0.5 * torch.randn(N, D, device="cuda", generator=g) - 1.0
    z = (mu[:M] if M <= N else mu.repeat((M + N - 1) // N, 1)[:M]) + torch.randn(M, D, device="cuda", generator=g) * torch.exp(0.5 * logvar[:1])
    path = int(os.environ.get("LVAE_PATH", "0"))        # 3: the experimental bf16x3 path (joint and prod-marginals)
    E.log_qz_raw(mu, logvar, z, math.log(N), True, path=path)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        E.log_qz_raw(mu, logvar, z, math.log(N), True, path=path)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"path={path} N={N} M={M} D={D} joint+pm: {dt*1e3:.2f} ms  {N*M*D/dt:.3e} (i,j,d) terms/s", flush=True)

if __name__ == "__main__":
    for a in sys.argv[1:]:
        N, M, D = (int(x) for x in a.split(","))
        run(N, M, D)
