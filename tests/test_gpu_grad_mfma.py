"""Backward of log q(z) on the matrix cores (pair_grad_mfma.hip behind lvae_log_qz_bwd_f32) against float64 PyTorch
autograd of the reference's formula (loss_and_optimisation.py:139-221), at sizes above the training-time sweeps, and
the goldens of test_gpu_grad.py through the same kernels.  Tolerance: 1e-4 of the largest gradient entry (the
north-star bar), stated per assertion."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _reference_grads(mu, lv, z, a, log_norm, chunk=256):
    mu64, lv64, z64 = (t.detach().double().requires_grad_(True) for t in (mu, lv, z))
    a64 = a.double()
    for lo in range(0, z.shape[0], chunk):
        zc = z64[lo:lo + chunk]
        dens = (-0.5 * (lv64[None] + math.log(2 * math.pi) + (zc[:, None, :] - mu64[None]) ** 2 * torch.exp(-lv64)[None])).sum(2)
        ((torch.logsumexp(dens, 1) - log_norm) * a64[lo:lo + chunk]).sum().backward()
    return mu64.grad, lv64.grad, z64.grad


def _our_grads(mu, lv, z, a):
    from lvae import estimators as E
    mu, lv, z = (t.detach().clone().requires_grad_(True) for t in (mu, lv, z))
    lq, _ = E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False)
    (a * lq).sum().backward()
    return mu.grad, lv.grad, z.grad


def _check(got, want, tol=1e-4):
    for name, g, w in zip(("d_mu", "d_logvar", "d_z"), got, want):
        scale = float(w.abs().max())
        err = float((g.double() - w).abs().max())
        assert err <= tol * scale, (name, err, scale)


def _problem(seed, M, N, D, sigma_scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mu = torch.randn(N, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(N, D, device="cuda", generator=g) - 1.0 + 2.0 * math.log(sigma_scale)
    src = torch.randint(0, N, (M,), device="cuda", generator=g)
    z = mu[src] + torch.randn(M, D, device="cuda", generator=g) * torch.exp(0.5 * lv[src])
    a = torch.randn(M, device="cuda", generator=g)
    return mu, lv, z, a


@pytest.mark.parametrize("M,N,D", [(2048, 3000, 32), (1500, 2500, 8), (3000, 2048, 48), (2100, 2000, 100),
                                   (2048, 2048, 128), (130, 70, 32), (4100, 5000, 64), (9000, 600, 33)])
def test_matrix_core_backward_vs_float64_autograd(M, N, D, lvae_env):
    lvae_env.set("LVAE_BWD_PATH", "mfma")
    mu, lv, z, a = _problem(M + D, M, N, D)
    _check(_our_grads(mu, lv, z, a), _reference_grads(mu, lv, z, a, math.log(N)))


@pytest.mark.parametrize("D,tol", [(31, 1e-4), (16, 1e-4), (64, 3e-4)])   # (the wider the latent, the farther the
def test_far_samples_keep_normalised_weights(D, tol, lvae_env):                        #  far samples: |v| ~ 2e4 at D = 64 costs d_z one ulp of it, 1.2e-4)
    """Samples far from every posterior (|v| ~ 1e4): their weights only come out right because sweep B evaluates v_ij
    bit-identically to sweep A, whose sums normalise them -- with the bf16-limb contraction (D <= 64) that needs the limb
    products accumulated in the same order in both sweeps (4e-4 on d_mu otherwise, 1e-6 with it)."""
    lvae_env.set("LVAE_BWD_PATH", "mfma")
    M, N = 1546, 653
    mu, lv, z, a = _problem(17, M, N, D, sigma_scale=0.3)
    z = z.clone()
    z[: M // 7] += 6.0
    _check(_our_grads(mu, lv, z, a), _reference_grads(mu, lv, z, a, math.log(N), chunk=64), tol=tol)


@pytest.mark.parametrize("case", ["b32", "ragged_d20", "d64", "n300", "d100", "d128"])
def test_matrix_core_backward_goldens(case, lvae_env):
    """The reference's own autograd gradients (tests/golden/grad_*.npz), joint term, through the MFMA kernels."""
    from lvae import estimators as E
    lvae_env.set("LVAE_BWD_PATH", "mfma")
    g = load_golden("grad_" + case)
    dev = lambda k, grad=False: torch.from_numpy(np.ascontiguousarray(g[k])).cuda().requires_grad_(grad)
    mu, lv, z = dev("mu", True), dev("logvar", True), dev("z", True)
    lq, _ = E.approximate_log_q_z(mu, lv, z, method="chen", dataset_size=10000, prod_marginals=False, reduce_mean=False)
    (dev("a") * lq).sum().backward()
    for name, got in (("mu", mu.grad), ("logvar", lv.grad), ("z", z.grad)):
        want = g[f"d_{name}_joint"]
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * float(np.abs(want).max()), err_msg=name)


def test_auto_path_selects_by_conditioning():
    """AUTO: above 2^22 pairs the matrix-core sweeps run behind the forward's conditioning gate; posteriors a
    thousand standard deviations from the centre fail it and the vector-ALU sweeps (exact differences) answer."""
    mu, lv, z, a = _problem(5, 2304, 2048, 32)
    _check(_our_grads(mu, lv, z, a), _reference_grads(mu, lv, z, a, math.log(2048)))
    mu, lv, z, a = _problem(6, 2304, 2048, 32, sigma_scale=1e-3)
    _check(_our_grads(mu, lv, z, a), _reference_grads(mu, lv, z, a, math.log(2048)))


def test_analysis_scale_wide_latent_needs_no_pair_matrix():
    """D = 96 at 20,000 x 20,000: the joint gradient runs on the matrix cores (no N x M buffer); the prod-marginals
    gradient of a wide latent still keeps the explicit matrix and is refused at this size."""
    from lvae import _native as N
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(9)
    n, D = 20000, 96
    mu = torch.randn(n, D, device="cuda", generator=g).requires_grad_(True)
    lv = (0.3 * torch.randn(n, D, device="cuda", generator=g) - 0.5)
    z = (mu.detach() + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv))
    lq, _ = E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False)
    lq.sum().backward()
    rows = torch.randperm(n, device="cuda", generator=g)[:16]          # d mu of 16 posteriors, float64, all samples
    m64, v64, z64 = mu.detach().double(), lv.double(), z.double()
    lse = torch.empty(n, dtype=torch.float64, device="cuda")
    for lo in range(0, n, 500):
        d = (-0.5 * (v64[None] + math.log(2 * math.pi) + (z64[lo:lo + 500, None, :] - m64[None]) ** 2 * torch.exp(-v64)[None])).sum(2)
        lse[lo:lo + 500] = torch.logsumexp(d, 1)
    dj = (-0.5 * (v64[rows][None] + math.log(2 * math.pi) + (z64[:, None, :] - m64[rows][None]) ** 2 * torch.exp(-v64[rows])[None])).sum(2)
    w = torch.exp(dj - lse[:, None])                                    # [n, 16]
    want = torch.einsum("ij,ijd->jd", w, (z64[:, None, :] - m64[rows][None]) * torch.exp(-v64[rows])[None])
    got = mu.grad[rows].double()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    big = torch.zeros(20000, 96, device="cuda", requires_grad=True)
    _, lqpm = E.approximate_log_q_z(big, torch.zeros_like(big), big.detach(), method="all", prod_marginals=True,
                                    reduce_mean=False)
    with pytest.raises(N.LvaeError, match="training batches"):
        lqpm.sum().backward()


@pytest.mark.parametrize("M,N,D", [(1, 1, 1), (5, 1, 3), (1, 7, 16), (33, 31, 17), (129, 257, 65)])
def test_matrix_core_backward_ragged_and_tiny_shapes(M, N, D, lvae_env):
    lvae_env.set("LVAE_BWD_PATH", "mfma")
    mu, lv, z, a = _problem(100 + M + N + D, M, N, D)
    _check(_our_grads(mu, lv, z, a), _reference_grads(mu, lv, z, a, math.log(N)), tol=2e-5)


def test_matrix_core_backward_partial_outputs_through_the_c_abi(lvae_env):
    """Each output gradient on its own (NULL for the others) equals its entry of the full call: sweep B is skipped
    without d_mu / d_logvar, sweep A always runs (it also yields the weight sums of the renormalisation)."""
    from lvae import _native as N
    from lvae import estimators as E
    lvae_env.set("LVAE_BWD_PATH", "mfma")
    M, Nn, D = 700, 900, 40
    mu, lv, z, a = _problem(77, M, Nn, D)
    lq, _ = E.log_qz_raw(mu, lv, z, math.log(Nn), False)
    nbytes = N._i64(0)
    N.check(N.lib().lvae_log_qz_bwd_workspace_bytes(M, Nn, D, 0, nbytes))
    ws = N.workspace(nbytes.value, z.device)

    def call(want_z, want_mu, want_lv):
        out = [torch.full_like(z, float("nan")) if want_z else None, torch.full_like(mu, float("nan")) if want_mu else None,
               torch.full_like(lv, float("nan")) if want_lv else None]
        N.check(N.lib().lvae_log_qz_bwd_f32(N.ptr(z), N.ptr(mu), N.ptr(lv), M, Nn, D, math.log(Nn), N.ptr(lq), N.ptr(a),
                                            None, N.ptr(out[0]), N.ptr(out[1]), N.ptr(out[2]), N.ptr(ws), nbytes.value,
                                            N.stream_ptr(z.device)))
        return out

    full = call(True, True, True)
    assert all(bool(torch.isfinite(t).all()) for t in full)
    for k in range(3):
        want = [False, False, False]
        want[k] = True
        part = call(*want)
        # (not bit for bit: the centring shift comes from float64 atomics whose order varies from call to call)
        assert float((part[k] - full[k]).abs().max()) <= 1e-5 * float(full[k].abs().max())
    # a workspace one byte short is refused before anything is launched
    with pytest.raises(N.LvaeError, match="workspace"):
        N.check(N.lib().lvae_log_qz_bwd_f32(N.ptr(z), N.ptr(mu), N.ptr(lv), M, Nn, D, math.log(Nn), N.ptr(lq), N.ptr(a),
                                            None, N.ptr(full[0]), None, None, N.ptr(ws), nbytes.value - 1,
                                            N.stream_ptr(z.device)))


def test_wide_latents_beyond_the_vector_alu_envelope_fail_loudly_when_ill_conditioned():
    """64 < D <= 128 with M * N > 2^28 has no vector-ALU branch (its explicit pair matrix is capped at 2^28 entries), so
    the matrix-core backward is the only form: well-conditioned posteriors give gradients within the bar of a float64
    spot check, and an estimate past the gate's limit yields NaN in every requested gradient instead of a silently
    degraded d logvar (include/lvae.h, lvae_log_qz_bwd_f32)."""
    M = N = 16400                       # 2.69e8 pairs > 2^28
    D = 100
    mu, lv, z, a = _problem(5, M, N, D)
    got = _our_grads(mu, lv, z, a)
    assert all(bool(torch.isfinite(t).all()) for t in got)
    # float64 spot check of d z on a few samples (the full reference would be 2.7e8 x 100 terms on the host path)
    rows = torch.tensor([0, 777, 8191, 16399], device="cuda")
    z64 = z[rows].double().requires_grad_(True)
    dens = (-0.5 * (lv.double()[None] + math.log(2 * math.pi) + (z64[:, None, :] - mu.double()[None]) ** 2
                    * torch.exp(-lv.double())[None])).sum(2)
    ((torch.logsumexp(dens, 1) - math.log(N)) * a[rows].double()).sum().backward()
    scale = float(z64.grad.abs().max())
    assert float((got[2][rows].double() - z64.grad).abs().max()) <= 1e-4 * scale
    # posteriors far from their common mean with small sigma: estimate >> 2e4 -> NaN, not a quiet loss of accuracy
    bad = _our_grads(mu * 40.0, lv - 4.0, z * 40.0, a)
    assert all(bool(torch.isnan(t).all()) for t in bad)
