"""GPU parity of the experimental opt-in split-bf16 pair kernel (LVAE_PATH_MFMA_BF16X3): same goldens, same
oracle and same tolerances as the fp32 MFMA path it mirrors (D <= 64, no diagonal exclusion)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
BX = 3


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


@pytest.mark.parametrize("case", ["generic", "ragged", "d64", "collapsed", "near_collapsed", "far", "d20", "d16", "n1"])
def test_bf16x3_golden(case):
    from lvae import estimators as E
    g = load_golden("logqz_" + case)
    mu, logvar, z = dev(g["mu"]), dev(g["logvar"]), dev(g["z"])
    for tag, method, ds in (("chen10000", "chen", 10000), ("all", "all", None)):
        lq, lqpm = E.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds, prod_marginals=True,
                                         reduce_mean=False, path=BX)
        ref, refpm = g[f"log_q_z_{tag}"], g[f"log_q_z_prod_marg_{tag}"]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale)
        np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=1e-4 * scale)
        assert rel(lq.double().mean(), ref.astype(np.float64).mean()) < 1e-4


@pytest.mark.parametrize("N,M,D", [(1000, 777, 32), (4096, 4096, 32), (513, 2049, 16), (200, 100, 8), (33, 1, 5),
                                    (513, 2049, 64), (3000, 4100, 48)])
def test_bf16x3_vs_oracle(N, M, D):
    from lvae import estimators as E
    rng = np.random.default_rng(N + M + D)
    mu = rng.standard_normal((N, D)).astype(np.float32)
    logvar = (0.5 * rng.standard_normal((N, D)) - 1.0).astype(np.float32)
    idx = np.arange(M) % N
    z = (mu[idx] + rng.standard_normal((M, D)) * np.exp(0.5 * logvar[idx])).astype(np.float32)
    ref, _ = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False)
    lq, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                  reduce_mean=False, path=BX)
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale)
    assert rel(lq.double().mean(), ref.mean()) < 1e-4


def test_bf16x3_redo_and_skip_branches():
    """Far samples underflow the fixed reference (redo with the running-maximum epilogue); heterogeneous
    posteriors switch the fixed-reference pass off altogether."""
    from lvae import estimators as E
    rng = np.random.default_rng(3)
    N_, D = 900, 32
    mu = rng.standard_normal((N_, D)).astype(np.float32)
    logvar = (0.3 * rng.standard_normal((N_, D)) - 0.5).astype(np.float32)
    z = (mu + rng.standard_normal((N_, D)) * np.exp(0.5 * logvar)).astype(np.float32)
    z[100:180] += 12.0
    ref, _ = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False)
    lq, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                  reduce_mean=False, path=BX)
    np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=2e-4, atol=2e-3)
    logvar[:5] = -14.0
    mu[:5] = 1e-3 * rng.standard_normal((5, D))
    z = (mu + rng.standard_normal((N_, D)) * np.exp(0.5 * logvar)).astype(np.float32)
    ref, _ = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False)
    lq, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                  reduce_mean=False, path=BX)
    np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=2e-4, atol=2e-3)


def test_bf16x3_prod_marginals_on_the_matrix_pipe():
    """prod-marginals under the opt-in path (D <= 64): the rank-3 polynomial per dimension as ONE bf16 MFMA per
    tile (pair_pm_bx.hip), with the same redo mechanism as the VALU kernel: far samples underflow the fixed
    reference and are redone by pair_pm_online_kernel; the other waves keep their fast-pass sums."""
    from lvae import estimators as E
    rng = np.random.default_rng(7)
    for N_, M_, D in ((700, 1000, 32), (333, 90, 16), (2100, 4200, 20), (600, 500, 64), (300, 400, 40)):
        mu = rng.standard_normal((N_, D)).astype(np.float32)
        logvar = (0.3 * rng.standard_normal((N_, D)) - 2.0).astype(np.float32)
        z = (mu[rng.integers(0, N_, M_)] + 0.4 * rng.standard_normal((M_, D))).astype(np.float32)
        z[M_ // 10: M_ // 10 + 40] += 40.0          # far in every dimension -> redo
        z[M_ // 2: M_ // 2 + 11, min(3, D - 1)] -= 25.0   # far in one dimension only
        ref, refpm = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=True, reduce_mean=False)
        lq, lqpm = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True,
                                         reduce_mean=False, path=BX)
        np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=2e-3, err_msg=str((N_, M_, D)))
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=2e-3, err_msg=str((N_, M_, D)))
        lq1, lqpm1 = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True,
                                           reduce_mean=False)
        assert float((lqpm - lqpm1).abs().max()) <= 2e-3


def test_bf16x3_prod_marginals_ill_conditioned_takes_the_direct_form():
    """Posteriors a few thousand standard deviations from the centre: the per-dimension polynomial cancels beyond
    what three limbs hold, the device-side conditioning test (the one AUTO uses) sends the prod-marginals sweep to
    the packed-fp32 direct form, and the opt-in path still matches the float64 oracle."""
    from lvae import estimators as E
    rng = np.random.default_rng(8)
    N_, M_, D = 600, 700, 32
    mu = (3.0 * rng.standard_normal((N_, D))).astype(np.float32)
    logvar = (0.3 * rng.standard_normal((N_, D)) - 14.0).astype(np.float32)        # sigma ~ 1e-3
    src = rng.integers(0, N_, M_)
    z = (mu[src] + rng.standard_normal((M_, D)) * np.exp(0.5 * logvar[src])).astype(np.float32)
    _, refpm = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=True, reduce_mean=False)
    lq, lqpm = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True, reduce_mean=False,
                                     path=BX)
    scale = float(np.abs(refpm).max())
    np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=1e-4 * scale)
    assert rel(lqpm.double().mean(), refpm.mean()) < 1e-4


@pytest.mark.parametrize("n,D", [(65536, 32), (32768, 64)])
def test_bf16x3_matches_fp32_path_at_full_size(n, D):
    """C2 (N = M = 65,536, D = 32) and a D = 64 case: per-sample agreement with the fp32 MFMA path."""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(9)
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(n, D, device="cuda", generator=g) - 1.0
    z = mu + torch.randn(n, D, device="cuda", generator=g) * torch.exp(0.5 * lv)
    a, _ = E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False, path=1)
    b, _ = E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=False, reduce_mean=False, path=BX)
    assert float((a - b).abs().max()) < 2e-4 * (D / 32)
    assert rel(b.double().mean(), a.double().mean()) < 1e-6


def test_bf16x3_rejects_what_it_does_not_cover():
    from lvae import _native as N
    from lvae import estimators as E
    mu = torch.zeros(64, 100, device="cuda")
    with pytest.raises(N.LvaeError, match="bf16x3"):
        E.approximate_log_q_z(mu, mu, mu, method="all", prod_marginals=False, reduce_mean=False, path=BX)
    x = torch.zeros(64, 200, device="cuda")          # kernel sums: the split-bf16 form stops at D = 128
    with pytest.raises(N.LvaeError, match="split-bf16"):
        E.kernel_sum(x, x, [0.5], 1e-5, path=BX)
    with pytest.raises(N.LvaeError, match="unknown path"):
        E.kernel_sum(x[:, :32].contiguous(), x[:, :32].contiguous(), [0.5], 1e-5, path=9)


def test_bf16x3_mi_bounds_with_excluded_diagonal():
    """The MBU bound masks the diagonal out of the logsumexp (mutual_information.py:207): same golden vectors,
    the split-bf16 kernel with the exclusion folded into the bias operand."""
    from lvae import mutual_information as MI
    for case in ("k128", "k16"):
        g = load_golden("mi_bounds_" + case)
        z, mulv = dev(g["z"]), dev(g["mu_logvars"])
        D = g["mu_logvars"].shape[1] // 2
        mu, std = mulv[:, :D].contiguous(), torch.exp(0.5 * mulv[:, D:]).contiguous()
        lower, upper = MI._bounds(z, (mu, std), path=BX).tolist()
        assert abs(lower - float(g["lower"])) <= 1e-4 * abs(float(g["lower"]))
        assert abs(upper - float(g["upper"])) <= 1e-4 * abs(float(g["upper"]))
    rng = np.random.default_rng(3)
    K, D = 1500, 32
    mu = rng.standard_normal((K, D)).astype(np.float32)
    lv = (0.5 * rng.standard_normal((K, D)) - 1).astype(np.float32)
    z = (mu + rng.standard_normal((K, D)).astype(np.float32) * np.exp(0.25 * lv)).astype(np.float32)
    lo, up = onp.calc_upper_and_lower_bound_representational_mi(z, np.concatenate([mu, lv], 1))
    glo, gup = MI._bounds(dev(z), (dev(mu), dev(np.exp(0.5 * lv))), path=BX).tolist()
    assert abs(glo - lo) <= 1e-4 * abs(lo) and abs(gup - up) <= 1e-4 * abs(up)


# ---- the kernel-sum / MMD entry points on the gated split-bf16 form ------------------------------------------
# The same tests as for the fp32 contraction (same oracle, same tolerances), run with LVAE_PATH_AUTO's size threshold
# for the split-bf16 kernel sums lowered to zero (LVAE_MMD_BX_MIN_PAIRS, a development switch): every call then takes the
# gated form -- both contractions prepared and launched, the device-side test picks the split-bf16 kernel.
import test_gpu_misc as TM     # noqa: E402


@pytest.fixture
def bf16x3_math(lvae_env):
    from lvae import _native as N
    lvae_env.set("LVAE_MMD_BX_MIN_PAIRS", "0")
    yield
    lvae_env.unset("LVAE_MMD_BX_MIN_PAIRS")


@pytest.mark.parametrize("case", ["n256", "n100_d64"])
def test_bf16x3_biased_mmd(case, bf16x3_math):
    TM.test_biased_mmd(case)


@pytest.mark.parametrize("n1,n2,D", [(300, 300, 32), (257, 190, 64), (1000, 1200, 32)])
def test_bf16x3_tts_mmd(n1, n2, D, bf16x3_math):
    TM.test_tts_mmd_unpinned(n1, n2, D)


@pytest.mark.parametrize("na,nb,D,off", [(300, 500, 32, 120), (257, 257, 64, 0), (90, 700, 16, -10), (130, 64, 128, 40)])
def test_bf16x3_kernel_sum_blocks(na, nb, D, off, bf16x3_math):
    TM.test_kernel_sum_blocks(na, nb, D, off)           # (D = 128 stays on the fp32 contraction)


@pytest.mark.parametrize("n1,n2,D,P", [(300, 340, 32, 20), (257, 190, 64, 130), (1000, 1500, 16, 9), (2048, 2048, 32, 33)])
def test_bf16x3_mmd_permutation_test(n1, n2, D, P, bf16x3_math):
    TM.test_mmd_permutation_test_unpinned(n1, n2, D, P)


def test_bf16x3_mmd_full_size_properties(bf16x3_math):
    TM.test_mmd_full_size_properties()


def test_bf16x3_mmd_is_a_different_kernel_with_the_same_answer():
    from lvae import _native as N
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(8192, 32, device="cuda", generator=g)
    y = torch.randn(8192, 32, device="cuda", generator=g) * 1.05
    alphas = [0.1 * i for i in range(15)]
    a = float(E.mmd_rbf(x, y, alphas, 1e-5, True, path=N.PATH_MFMA)[0])
    b = float(E.mmd_rbf(x, y, alphas, 1e-5, True, path=BX)[0])
    assert abs(a - b) <= 2e-6 * 15
    assert a != b or True       # (different kernels; equality to the last bit is neither expected nor required)


def test_auto_kernel_sums_gate(lvae_env):
    """LVAE_PATH_AUTO on the kernel sums: above the size threshold both contractions are launched and a device-side
    test on the largest squared norm lets one run.  Well-conditioned points: bit-equal to the forced split-bf16
    path; points far from the origin (alpha_max * 4 max|p|^2 > 2e4): bit-equal to the forced fp32 path; below the
    threshold: the fp32 path.  Every outcome within the oracle's tolerance that its conditioning allows."""
    from lvae import _native as N
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(12)
    a = torch.randn(700, 32, device="cuda", generator=g)
    b = torch.randn(900, 32, device="cuda", generator=g)
    alphas = [0.1 * i for i in range(15)]
    small = float(E.kernel_sum(a, b, alphas, 1e-5))
    assert small == float(E.kernel_sum(a, b, alphas, 1e-5, path=N.PATH_MFMA))       # 6.3e5 pairs: below the threshold
    lvae_env.set("LVAE_MMD_BX_MIN_PAIRS", "0")
    try:
        for shift, same_as in ((0.0, BX), (20.0, N.PATH_MFMA)):       # |p|^2 ~ 32 / ~ 12,800: 1.4 * 4 * that <> 2e4
            aa, bb = a + shift, b + shift
            got = float(E.kernel_sum(aa, bb, alphas, 1e-5))
            # (the same kernel on the same operands; the float64 column sums are merged with atomics, so the total may
            #  differ in its last bit -- the other contraction differs at 1e-8)
            assert abs(got - float(E.kernel_sum(aa, bb, alphas, 1e-5, path=same_as))) <= 1e-13 * abs(got), shift
            want = float(E.kernel_sum(aa, bb, alphas, 1e-5, path=N.PATH_DIRECT))    # exact differences
            assert abs(got - want) <= (1e-6 if shift == 0.0 else 2e-3) * abs(want), (shift, got, want)
        # the symmetric walk, the excluded diagonal and the permutation sums take the same gate
        pooled = torch.cat((a, b), 0).contiguous()
        got = float(E.kernel_sum(pooled[:704], pooled, alphas, 1e-5, True, 0, symmetric=True))
        assert got == float(E.kernel_sum(pooled[:704], pooled, alphas, 1e-5, True, 0, symmetric=True, path=BX))
        labels = E.permutation_labels(700, 900, 9, pooled.device, torch.Generator(device="cuda").manual_seed(1))
        for shift, same_as in ((0.0, BX), (20.0, N.PATH_MFMA)):
            pp = pooled + shift
            q, lr, tot = E.mmd_permutation_sums(pp, 0, pp, labels, alphas, 1e-5)
            q2, lr2, tot2 = E.mmd_permutation_sums(pp, 0, pp, labels, alphas, 1e-5, path=same_as)
            assert torch.equal(q, q2) and torch.equal(lr, lr2) and float(tot) == float(tot2), shift
        with pytest.raises(N.LvaeError, match="direct-difference"):
            E.mmd_permutation_sums(pooled, 0, pooled, labels, alphas, 1e-5, path=N.PATH_DIRECT)
    finally:
        lvae_env.unset("LVAE_MMD_BX_MIN_PAIRS")
