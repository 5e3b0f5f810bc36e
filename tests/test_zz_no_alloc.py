"""The allocation-free entry point lvae_log_qz_noalloc_f32 (static device scratch inside the library) against
lvae_log_qz_f32 (caller-provided workspace): same kernels, so the outputs must agree to the parity bar (1e-4
relative on the means, and element-wise to float32 round-off), on two calls in a row, the second smaller --
the shared scratch keeps the first call's contents -- and it must refuse shapes beyond its fixed scratch."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _inputs(seed, M, Nn, D):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((Nn, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((Nn, D)) - 1.0).astype(np.float32)
    z = (mu[rng.integers(0, Nn, M)] + rng.standard_normal((M, D)) * 0.6).astype(np.float32)
    return mu, lv, z


def _both(mu, lv, z, pm):
    from lvae import _native as N
    lib = N.lib()
    tmu, tlv, tz = (torch.from_numpy(a).cuda() for a in (mu, lv, z))
    Nn, D = mu.shape
    M = z.shape[0]
    log_norm = math.log(Nn)
    st = N.stream_ptr(tz.device)
    nbytes = ctypes.c_int64(0)
    N.check(lib.lvae_log_qz_workspace_bytes(M, Nn, D, int(pm), nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
    ref = torch.empty(M, device="cuda")
    ref_pm = torch.empty(M, device="cuda") if pm else None
    N.check(lib.lvae_log_qz_f32(N.ptr(tz), N.ptr(tmu), N.ptr(tlv), 0, M, Nn, D, log_norm, 0, 0, 0, N.ptr(ref),
                                N.ptr(ref_pm), N.ptr(ws), nbytes.value, st))
    out = torch.full((M,), float("nan"), device="cuda")
    out_pm = torch.full((M,), float("nan"), device="cuda") if pm else None
    # one warm call first: the first call on a device looks the scratch array's address up (hipGetSymbolAddress) and
    # loads the kernels' code objects; what must not allocate is every call after it
    N.check(lib.lvae_log_qz_noalloc_f32(N.ptr(tz), N.ptr(tmu), N.ptr(tlv), 0, M, Nn, D, log_norm, 0, 0, 0,
                                        N.ptr(out), N.ptr(out_pm), st))
    torch.cuda.synchronize()
    out.fill_(float("nan"))
    if pm:
        out_pm.fill_(float("nan"))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()          # torch's caching allocator ...
    free_before = torch.cuda.mem_get_info()[0]      # ... and the driver's own view: a hipMalloc inside liblvae.so shows here
    N.check(lib.lvae_log_qz_noalloc_f32(N.ptr(tz), N.ptr(tmu), N.ptr(tlv), 0, M, Nn, D, log_norm, 0, 0, 0,
                                        N.ptr(out), N.ptr(out_pm), st))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] >= free_before, "device memory was allocated inside the allocation-free call"
    return ref, ref_pm, out, out_pm


def _check(ref, got, want64):
    r, g = ref.cpu().numpy().astype(np.float64), got.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    np.testing.assert_allclose(g, r, rtol=1e-6, atol=1e-5)            # same kernels, same plan
    assert abs(g.mean() - want64.mean()) <= 1e-4 * abs(want64.mean())  # and the oracle, to the parity bar
    np.testing.assert_allclose(g, want64, rtol=1e-4, atol=5e-3)


def test_noalloc_matches_workspace_entry_point_twice_second_smaller():
    # first call: a larger problem with prod-marginals (fills most regions of the scratch)
    mu, lv, z = _inputs(0, 1500, 2000, 32)
    ref, ref_pm, out, out_pm = _both(mu, lv, z, True)
    want, want_pm = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    _check(ref, out, want)
    _check(ref_pm, out_pm, want_pm)
    # second call, smaller and ragged, on the scratch the first call left behind
    mu, lv, z = _inputs(1, 77, 100, 20)
    ref, ref_pm, out, out_pm = _both(mu, lv, z, True)
    want, want_pm = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    _check(ref, out, want)
    _check(ref_pm, out_pm, want_pm)
    # third: joint density only, the far-away regime (redo pass reads flags that must have been cleared)
    mu, lv, z = _inputs(2, 300, 260, 32)
    z[:40] += 30.0
    ref, _, out, _ = _both(mu, lv, z, False)
    want, _ = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False)
    _check(ref, out, want)


def test_noalloc_rejects_shapes_beyond_its_scratch():
    from lvae import _native as N
    lib = N.lib()
    nbytes = ctypes.c_int64(0)
    N.check(lib.lvae_log_qz_workspace_bytes(65536, 65536, 32, 0, nbytes))
    assert nbytes.value > N.NOALLOC_SCRATCH_BYTES
    d = torch.zeros(8, device="cuda")
    rc = lib.lvae_log_qz_noalloc_f32(N.ptr(d), N.ptr(d), N.ptr(d), 0, 65536, 65536, 32, 0.0, 0, 0, 0, N.ptr(d), None,
                                     N.stream_ptr(d.device))
    assert rc == -4 and b"scratch" in lib.lvae_last_error()
