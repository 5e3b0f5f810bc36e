"""Whole-call latency of the marginal-KL / MI statistics at small sizes (BASELINE configs[0]: N = 4,096),
eager launches vs one HIP-graph replay."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import distributed as LD

def run(n, D=32):
    g = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    f = lambda: LD.hip_local_stats(mu, lv, z, mu, lv, math.log(n), False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e6
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sums, lq = f()
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 50 * 1e6
    s = sums.tolist()
    print(f"N=M={n} D={D}: eager {eager:.0f} us, graph replay {rep:.0f} us  ({n*n/rep*1e6:.3e} pairs/s; MFMA floor {n*n*4*D/157.3e12*1e6:.0f} us)  KL={(s[0]-s[2])/s[4]:.4f}", flush=True)

for n in (256, 1024, 4096, 16384):
    run(n)
