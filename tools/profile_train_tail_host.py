"""Host-side profile (cProfile) of the training-time tail at B=32: where the ~0.8 ms per step goes."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_train_tail as T   # runs its own table first

B = 32
g = torch.Generator(device="cuda").manual_seed(B)
mu = torch.randn(B, 32, device="cuda", generator=g).requires_grad_(True)
logvar = (0.5 * torch.randn(B, 32, device="cuda", generator=g) - 1).requires_grad_(True)
eps = torch.randn(B, 1, 32, device="cuda", generator=g)
for _ in range(20):
    mu.grad = logvar.grad = None
    T.hip_tail(mu, logvar, eps).backward()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    mu.grad = logvar.grad = None
    T.hip_tail(mu, logvar, eps).backward()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
