"""One rank's compute of the weak-scaling bench on ONE GPU: the single call (all posteriors after a blocking
all-gather) against the two-block form of the overlapped exchange (own posteriors, then the others, merged)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E

def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

D = 32
for world in (2, 4, 8):
    q = 256 * world
    n = max(q, int(round(65536 * math.sqrt(world) / q)) * q)
    nr = n // world
    g = torch.Generator(device="cuda").manual_seed(world)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    packed = torch.cat([mu, lv], 1).contiguous()
    rank = world // 2
    lo, hi = rank * nr, (rank + 1) * nr
    z = (mu[lo:hi] + torch.randn(nr, D, device="cuda", generator=g) * torch.exp(0.5 * lv[lo:hi])).contiguous()
    mu_l, lv_l = mu[lo:hi].contiguous(), lv[lo:hi].contiguous()

    def single():
        return E.log_qz_raw(packed[:, :D].contiguous(), packed[:, D:].contiguous(), z, math.log(n), False)[0]

    def two_block():
        own = E.log_qz_partial(mu_l, lv_l, z)
        others = torch.cat([packed[:lo], packed[hi:]], 0)
        rest = E.log_qz_partial(others[:, :D].contiguous(), others[:, D:].contiguous(), z)
        return E.lse_combine(own, rest, math.log(n))

    def own_only():
        return E.log_qz_partial(mu_l, lv_l, z)

    a, b = single(), two_block()
    t1, t2, t3 = timed(single), timed(two_block), timed(own_only)
    print(f"world={world} n_total={n} rows/rank={nr}: single call {t1:.3f} ms, two-block {t2:.3f} ms (+{(t2/t1-1)*100:.1f}%), "
          f"own block alone {t3:.3f} ms (the window that hides the all-gather); max |diff| {float((a-b).abs().max()):.2e}", flush=True)
