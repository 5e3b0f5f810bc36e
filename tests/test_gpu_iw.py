"""GPU parity of the per-sample densities, the reparameterised draw and the IW reduction against the
reference's golden vectors (tests/golden/iw_*.npz, logqz_*.npz) and the C oracle at a larger size."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("case", ["k1", "k600", "k37_d64"])
def test_reparam_iw_golden(case):
    from lvae import estimators as E
    g = load_golden("iw_" + case)
    out = E.reparam_iw(dev(g["mu"]), dev(g["logvar"]), dev(g["eps"]), dev(g["log_p_x_z"]))
    # z: same float32 multiply-then-add as the reference; exp() may differ by an ulp
    np.testing.assert_allclose(out["latent_z"].cpu().numpy(), g["z"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["log_q_z_x"].cpu().numpy(), g["log_q_z_x"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(out["log_p_z"].cpu().numpy(), g["log_p_z"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(out["iw_ll"].cpu().numpy(), g["iw_ll"], rtol=1e-4, atol=1e-3)
    ref_mean = g["iw_ll"].astype(np.float64).mean()
    assert abs(out["iw_ll"].double().mean().item() - ref_mean) / abs(ref_mean) < 1e-4
    # the stand-alone pieces agree with the fused pass
    lq = E.sample_log_likelihood(dev(g["z"]), dev(g["mu"]), dev(g["logvar"]), reduce_batch_dim=False)
    lp = E.sample_log_likelihood(dev(g["z"]), reduce_batch_dim=False)
    np.testing.assert_allclose(lq.cpu().numpy(), g["log_q_z_x"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(lp.cpu().numpy(), g["log_p_z"], rtol=1e-4, atol=1e-3)
    iw = E.iw_log_p_x(dev(g["log_p_x_z"]), dev(g["log_p_z"]), dev(g["log_q_z_x"]))
    np.testing.assert_allclose(iw.cpu().numpy(), g["iw_ll"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("case", ["generic", "d64", "d20", "peaked", "far"])
def test_sample_log_likelihood_golden(case):
    from lvae import estimators as E
    g = load_golden("logqz_" + case)
    if "log_q_z_x" not in g:
        pytest.skip("M != N")
    z, mu, lv = dev(g["z"]), dev(g["mu"]), dev(g["logvar"])
    tol = dict(rtol=1e-4, atol=1e-3)
    if case == "far":
        tol = dict(rtol=1e-4, atol=0.5)     # values ~ -1e4: float32 spacing
    np.testing.assert_allclose(E.sample_log_likelihood(z, mu, lv, reduce_batch_dim=False).cpu().numpy(),
                               g["log_q_z_x"], **tol)
    np.testing.assert_allclose(E.sample_log_likelihood(z, reduce_batch_dim=False).cpu().numpy(),
                               g["log_p_z"], **tol)
    np.testing.assert_allclose(E.sample_log_likelihood(z, mu, lv, False, False).cpu().numpy(),
                               g["log_q_z_x_nored"], **tol)
    np.testing.assert_allclose(E.sample_log_likelihood(z, None, None, False, False).cpu().numpy(),
                               g["log_p_z_nored"], **tol)
    got = float(E.sample_log_likelihood(z))
    assert abs(got - float(g["log_p_z_batchmean"])) <= 1e-4 * abs(float(g["log_p_z_batchmean"]))


def test_reparam_iw_vs_c_oracle_large():
    from lvae import estimators as E
    from oracle import oracle_c
    rng = np.random.default_rng(5)
    B, K, D = 300, 1000, 32
    mu = rng.standard_normal((B, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((B, D)) - 1).astype(np.float32)
    eps = rng.standard_normal((B, K, D)).astype(np.float32)
    lpxz = (5 * rng.standard_normal((B, K)) - 100).astype(np.float32)
    z, lq, lp, iw = oracle_c.reparam_iw(mu, lv, eps, lpxz)
    out = E.reparam_iw(dev(mu), dev(lv), dev(eps), dev(lpxz))
    np.testing.assert_allclose(out["latent_z"].cpu().numpy(), z, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["log_q_z_x"].cpu().numpy(), lq, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(out["log_p_z"].cpu().numpy(), lp, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(out["iw_ll"].cpu().numpy(), iw, rtol=1e-4, atol=1e-3)
    assert abs(out["iw_ll"].double().mean().item() - iw.mean()) / abs(iw.mean()) < 1e-4
    # K = 1 known answer: the estimate is the single log-weight
    out1 = E.reparam_iw(dev(mu), dev(lv), dev(eps[:, :1]), dev(lpxz[:, :1]))
    w = lpxz[:, 0] + lp[:, 0] - lq[:, 0]
    np.testing.assert_allclose(out1["iw_ll"].cpu().numpy(), w, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("D", [4, 8, 16, 32, 64, 128, 256, 20, 48, 100, 30, 7, 300, 768, 1030])
def test_reparam_iw_row_geometries(D):
    """Every lane-group width of the row reduction (G = D/4 = 1 ... 64 lanes: quad permutations, half-row mirror,
    row rotation, cross-row shuffles) and both output routes -- [B,K] rows staged through LDS (K a multiple of 4 from
    256) and written piecewise (other K) -- against the C oracle; and the latent sizes outside that kernel: short rows
    on the next power of two of lanes (D = 20, 48, 100 by 16-byte pieces; 30, 7 by elements), long ones on a wave per
    row (300, 768), beyond 1,024 one lane per row (1030)."""
    from lvae import estimators as E
    from oracle import oracle_c
    rng = np.random.default_rng(D)
    for B, K in ((5, 256), (3, 1000), (4, 259), (7, 100), (2, 2052), (9, 7)):
        mu = rng.standard_normal((B, D)).astype(np.float32)
        lv = (0.5 * rng.standard_normal((B, D)) - 1).astype(np.float32)
        eps = rng.standard_normal((B, K, D)).astype(np.float32)
        lpxz = (5 * rng.standard_normal((B, K)) - 100).astype(np.float32)
        z, lq, lp, iw = oracle_c.reparam_iw(mu, lv, eps, lpxz)
        for return_z in (False, True):
            out = E.reparam_iw(dev(mu), dev(lv), dev(eps), dev(lpxz), return_z=return_z)
            if return_z:
                np.testing.assert_allclose(out["latent_z"].cpu().numpy(), z, rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(out["log_q_z_x"].cpu().numpy(), lq, rtol=1e-4, atol=1e-3, err_msg=f"K={K}")
            np.testing.assert_allclose(out["log_p_z"].cpu().numpy(), lp, rtol=1e-4, atol=1e-3, err_msg=f"K={K}")
            np.testing.assert_allclose(out["iw_ll"].cpu().numpy(), iw, rtol=1e-4, atol=1e-3, err_msg=f"K={K}")
        only = E.reparam_iw(dev(mu), dev(lv), dev(eps), dev(lpxz), return_z=False, return_densities=False)
        assert only["log_q_z_x"] is None and only["log_p_z"] is None and only["latent_z"] is None
        np.testing.assert_allclose(only["iw_ll"].cpu().numpy(), iw, rtol=1e-4, atol=1e-3)
        # densities of given z (no reparameterisation): the same row reduction with z as the source
        lq2 = E.sample_log_likelihood(dev(z), dev(mu), dev(lv), reduce_batch_dim=False)
        np.testing.assert_allclose(lq2.cpu().numpy(), lq, rtol=1e-4, atol=1e-3)


def test_iw_full_size_properties():
    """BASELINE configs[4] at full size (K = 1,000 samples x 16,384 sentences, D = 32), through properties:
    a constant added to log p(x|z) shifts the IW estimate by that constant, the estimate lies between the
    mean and the maximum of the log-weights, and the per-sample densities agree with sample_log_likelihood."""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(2)
    B, K, D = 16384, 1000, 32
    mu = torch.randn(B, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(B, D, device="cuda", generator=g) - 1.0
    eps = torch.randn(B, K, D, device="cuda", generator=g)
    lpxz = 5.0 * torch.randn(B, K, device="cuda", generator=g) - 100.0
    a = E.reparam_iw(mu, lv, eps, lpxz, return_z=True)
    b = E.reparam_iw(mu, lv, eps, lpxz + 7.5, return_z=False)
    assert float((b["iw_ll"] - a["iw_ll"] - 7.5).abs().max()) <= 1e-4
    w = lpxz + a["log_p_z"] - a["log_q_z_x"]
    assert bool((a["iw_ll"] <= w.max(dim=1).values + 1e-4).all())
    assert bool((a["iw_ll"] >= w.mean(dim=1) - 1e-3).all())
    lq = E.sample_log_likelihood(a["latent_z"][:256], mu[:256], lv[:256], reduce_batch_dim=False)
    assert float((lq - a["log_q_z_x"][:256]).abs().max()) <= 1e-3
