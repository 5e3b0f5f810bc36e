"""Pair-kernel time (HIP events around the kernel, lvae_profile_*) against the number of reduced-axis splits,
forced through LVAE_NSPLIT: sweep_nsplit.py n D [ns ...]"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import _native as N, distributed as LD
n, D = int(sys.argv[1]), int(sys.argv[2])
splits = [int(a) for a in sys.argv[3:]] or [0, 4, 8, 16, 32, 64]
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(n, D, device="cuda", generator=g)
lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
for ns in splits:
    if ns:
        os.environ["LVAE_NSPLIT"] = str(ns)
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
    else:
        os.environ.pop("LVAE_NSPLIT", None)
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
    out = (ctypes.c_int64 * 8)()
    N.check(N.lib().lvae_plan_info(n, n, D, 0, out))
    for _ in range(5):
        LD.hip_local_stats(mu, lv, z, mu, lv, math.log(n), False)
    torch.cuda.synchronize()
    N.check(N.lib().lvae_profile_begin(64))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        sums, _ = LD.hip_local_stats(mu, lv, z, mu, lv, math.log(n), False)
    b.record()
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
    N.check(N.lib().lvae_profile_end(ctypes.byref(ms), ctypes.byref(cnt)))
    print(f"n={n} D={D} LVAE_NSPLIT={ns or 'auto'} -> nsplit {out[3]} x {out[4]} blocks, {out[2] // (128 * out[1])} column groups: "
          f"pair kernels {ms.value / cnt.value * 1e3:.1f} us, whole call {a.elapsed_time(b) / 20 * 1e3:.1f} us", flush=True)
