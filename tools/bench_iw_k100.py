import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lvae import _native as N
B, K, D = 65536, 100, 32
g = torch.Generator(device="cuda").manual_seed(0)
mu = torch.randn(B, D, device="cuda", generator=g); lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1
eps = torch.randn(B, K, D, device="cuda", generator=g); lpxz = 5 * torch.randn(B, K, device="cuda", generator=g) - 100
lq = torch.empty(B, K, device="cuda"); lp = torch.empty(B, K, device="cuda"); iw = torch.empty(B, device="cuda")
lib = N.lib(); st = N.stream_ptr(eps.device)
def call():
    N.check(lib.lvae_reparam_iw_f32(N.ptr(mu), N.ptr(lv), N.ptr(eps), N.ptr(lpxz), B, K, D, None, N.ptr(lq), N.ptr(lp), N.ptr(iw), st))
for cfg in (None, ("0","8"), ("0","4"), ("0","2"), ("1","4")):
    if cfg: os.environ["LVAE_IW_PIPE"], os.environ["LVAE_IW_U"] = cfg; __import__("lvae._native", fromlist=["reload_env"]).reload_env()
    for _ in range(3): call()
    torch.cuda.synchronize(); ts=[]
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); call(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(cfg, f"{np.median(ts):.3f} ms", flush=True)
