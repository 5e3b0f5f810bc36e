"""Backward of log q(z) (joint term, all three gradients) at analysis scale: matrix-core sweeps vs the vector-ALU
sweeps, per call, with the useful-flop rate of the matrix-core form (16 * M * N * D flop: two sweeps, each a
density contraction and a feature contraction of depth 2D) against the fp32 MFMA peak.
usage: python tools/bench_bwd.py [M,N,D ...]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvae import _native as N

PEAK = 157.3e12


def bwd(z, mu, lv, log_norm, lq, a):
    M, D = z.shape
    Nn = mu.shape[0]
    nbytes = N._i64(0)
    N.check(N.lib().lvae_log_qz_bwd_workspace_bytes(M, Nn, D, 0, nbytes))
    ws = N.workspace(nbytes.value, z.device)
    d_z, d_mu, d_lv = torch.empty_like(z), torch.empty_like(mu), torch.empty_like(lv)
    N.check(N.lib().lvae_log_qz_bwd_f32(N.ptr(z), N.ptr(mu), N.ptr(lv), M, Nn, D, log_norm, N.ptr(lq), N.ptr(a), None,
                                        N.ptr(d_z), N.ptr(d_mu), N.ptr(d_lv), N.ptr(ws), nbytes.value,
                                        N.stream_ptr(z.device)))
    return d_mu, d_lv, d_z


def run(M, Nn, D, reps=5):
    g = torch.Generator(device="cuda").manual_seed(1)
    mu = torch.randn(Nn, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(Nn, D, device="cuda", generator=g) - 1.0
    z = mu[torch.randint(0, Nn, (M,), device="cuda", generator=g)] + 0.5 * torch.randn(M, D, device="cuda", generator=g)
    a = torch.randn(M, device="cuda", generator=g)
    from lvae import estimators as E
    lq, _ = E.log_qz_raw(mu, lv, z, math.log(Nn), False)
    out = {}
    for path in ("mfma", "valu"):
        if path == "valu" and D > 64 and M * Nn > (1 << 28):
            continue
        os.environ["LVAE_BWD_PATH"] = path
        __import__("lvae._native", fromlist=["reload_env"]).reload_env()    # (switches are read once per process)
        grads = bwd(z, mu, lv, math.log(Nn), lq, a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            bwd(z, mu, lv, math.log(Nn), lq, a)
        torch.cuda.synchronize()
        out[path] = ((time.perf_counter() - t0) / reps, grads)
    ms = out["mfma"][0] * 1e3
    rate = 16.0 * M * Nn * D / out["mfma"][0]
    line = f"bwd M={M} N={Nn} D={D}: mfma {ms:.3f} ms ({rate / 1e12:.1f} TFLOP/s useful, {100 * rate / PEAK:.1f}% of fp32 MFMA peak)"
    if "valu" in out:
        dev = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(out["mfma"][1], out["valu"][1]))
        line += f"; valu {out['valu'][0] * 1e3:.3f} ms; max rel diff {dev:.2e}"
    print(line, flush=True)


if __name__ == "__main__":
    shapes = sys.argv[1:] or ["16384,16384,32", "16384,16384,64", "16384,16384,128", "4096,4096,32", "65536,65536,32"]
    for s in shapes:
        run(*[int(v) for v in s.split(",")])
