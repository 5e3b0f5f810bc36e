"""tts MMD statistic time against a forced number of reduced-axis splits (LVAE_NSPLIT applies to all three kernel
matrices): sweep_mmd_nsplit.py n D [ns ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lvae import two_sample as TS
n, D = int(sys.argv[1]), int(sys.argv[2])
splits = [int(a) for a in sys.argv[3:]] or [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, D, device="cuda", generator=g)
y = torch.randn(n, D, device="cuda", generator=g) * 1.05
alphas = [0.1 * i for i in range(15)]
stat = TS.MMDStatistic(n, n)
for ns in splits:
    if ns:
        os.environ["LVAE_NSPLIT"] = str(ns)
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
    else:
        os.environ.pop("LVAE_NSPLIT", None)
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
    for _ in range(3):
        v = stat(x, y, alphas)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); v = stat(x, y, alphas); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"n={n} D={D} LVAE_NSPLIT={ns or 'auto'}: median {np.median(ts) * 1e3:.0f} us  min {min(ts) * 1e3:.0f} us  stat {float(v):.6e}", flush=True)
