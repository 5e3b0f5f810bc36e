"""One MMD workload, a few times, for rocprofv3 passes: run_mmd.py {tts|biased|perm} n D [iters] [P]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import estimators as E
mode, n, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
P = int(sys.argv[5]) if len(sys.argv) > 5 else 100
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, D, device="cuda", generator=g)
y = 0.98 * torch.randn(n, D, device="cuda", generator=g) + 0.01
alphas = [0.1 * i for i in range(15)]
labels = E.permutation_labels(n, n, P, "cuda", g) if mode == "perm" else None
for _ in range(iters):
    if mode == "tts":
        out = E.mmd_rbf(x, y, alphas, 1e-5, unbiased=True)
    elif mode == "biased":
        out = E.mmd_rbf(x, y, [1.0 / D ** 2], 0.0, unbiased=False)
    else:
        out = E.mmd_permutation_test(x, y, alphas, labels=labels)["statistic"]
torch.cuda.synchronize()
print(mode, n, D, float(out.reshape(-1)[0]), flush=True)
