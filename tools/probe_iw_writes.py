"""What the two [B,K] density outputs of the IW call cost, by shape of the output rows: K = 1000 (rows of 4000 B: not a
multiple of the 128-B line), 1024 and 992 (multiples), one output matrix instead of two, against the reduction alone.
usage: python tools/probe_iw_writes.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import _native as N

def run(B, K, D, lq, lp, reps=30):
    g = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.randn(B, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1.0
    eps = torch.randn(B, K, D, device="cuda", generator=g)
    px = 5.0 * torch.randn(B, K, device="cuda", generator=g) - 100.0
    oq = torch.empty(B, K, device="cuda") if lq else None
    op = torch.empty(B, K, device="cuda") if lp else None
    iw = torch.empty(B, device="cuda")
    st = N.stream_ptr(eps.device)
    def call():
        N.check(N.lib().lvae_reparam_iw_f32(N.ptr(mu), N.ptr(lv), N.ptr(eps), N.ptr(px), B, K, D, None, N.ptr(oq), N.ptr(op),
                                            N.ptr(iw), st))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    N.check(N.lib().lvae_profile_begin(reps))
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    a, b = ctypes.c_double(0), ctypes.c_int(0)
    N.check(N.lib().lvae_profile_end(ctypes.byref(a), ctypes.byref(b)))
    ms = a.value / b.value
    by = B * K * (4 * D + 4 + 4 * (int(lq) + int(lp)))
    print(f"B={B} K={K} D={D} outputs lq={int(lq)} lp={int(lp)}: {ms:.4f} ms  {by / ms / 1e6:.0f} GB/s ({by / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)

if __name__ == "__main__":
    for K in (1000, 1024, 992):
        B = 16384 * 1000 // K
        for lq, lp in ((0, 0), (1, 1), (1, 0)):
            run(B, K, 32, lq, lp)
