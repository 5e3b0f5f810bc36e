"""GPU parity: approximate_log_q_z (HIP, through the C ABI) against the golden vectors produced by the
reference and against the NumPy oracle.  Tolerance: BASELINE north_star 1e-4 relative on the final
(mean) values; per-sample values get an absolute slack that covers fp32 rounding of O(100) numbers."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu

CASES = ["generic", "ragged", "d64", "collapsed", "near_collapsed", "peaked", "far", "d20", "d16", "n1"]
METHODS = [("chen10000", "chen", 10000), ("chen42068", "chen", 42068), ("all", "all", None)]
PATHS = {"auto": 0, "mfma": 1, "direct": 2}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("path", ["auto", "mfma", "direct"])
def test_log_qz_golden(case, path):
    from lvae import estimators as E
    g = load_golden("logqz_" + case)
    if path == "mfma" and case in ("peaked",):
        pytest.skip("the forced MFMA expansion is ill-conditioned here by design; auto picks direct")
    mu, logvar, z = dev(g["mu"]), dev(g["logvar"]), dev(g["z"])
    for tag, method, ds in METHODS:
        lq, lqpm = E.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds,
                                         prod_marginals=True, reduce_mean=False, path=PATHS[path])
        ref, refpm = g[f"log_q_z_{tag}"], g[f"log_q_z_prod_marg_{tag}"]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale)
        np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=1e-4 * scale)
        assert rel(lq.double().mean(), ref.astype(np.float64).mean()) < 1e-4
        assert rel(lqpm.double().mean(), refpm.astype(np.float64).mean()) < 1e-4
        lq2, none = E.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds,
                                          prod_marginals=False, reduce_mean=False, path=PATHS[path])
        assert none is None
        np.testing.assert_allclose(lq2.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale)
        a, b = E.approximate_log_q_z(mu, logvar, z, method=method, dataset_size=ds, prod_marginals=True,
                                     reduce_mean=True, path=PATHS[path])
        assert rel(a, g[f"log_q_z_mean_{tag}"]) < 1e-4 and rel(b, g[f"log_q_z_prod_marg_mean_{tag}"]) < 1e-4


@pytest.mark.parametrize("N,M,D", [(1000, 777, 32), (4096, 4096, 32), (513, 2049, 64), (300, 300, 128),
                                    (200, 100, 8), (150, 90, 48), (64, 64, 200),
                                    # wide latents (the blocked direct kernel: batches of 32 posteriors x 16 dimensions;
                                    # D % 4 != 0 takes its scalar row loads, N % 32 != 0 a ragged last batch)
                                    (300, 257, 768), (97, 300, 201), (33, 513, 130), (1100, 70, 256)])
def test_log_qz_vs_oracle(N, M, D):
    from lvae import estimators as E
    rng = np.random.default_rng(N + M + D)
    mu = rng.standard_normal((N, D)).astype(np.float32)
    logvar = (0.5 * rng.standard_normal((N, D)) - 1.0).astype(np.float32)
    eps = rng.standard_normal((M, D)).astype(np.float32)
    idx = np.arange(M) % N
    z = (mu[idx] + eps * np.exp(0.5 * logvar[idx])).astype(np.float32)
    ref, refpm = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=True, reduce_mean=False)
    for path in ("auto", "mfma", "direct"):
        if path == "mfma" and D > 128:
            continue
        lq, lqpm = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True,
                                         reduce_mean=False, path=PATHS[path])
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale, err_msg=path)
        np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=1e-4 * scale, err_msg=path)
        assert rel(lq.double().mean(), ref.mean()) < 1e-4, path
        assert rel(lqpm.double().mean(), refpm.mean()) < 1e-4, path


def test_prod_marginals_redo_path():
    """Samples ~50 sigma away from every posterior underflow the fixed-reference fast pass of the
    prod-marginals kernel; the flagged waves must be redone with the running-maximum form, and the
    other waves of the same call must keep their fast-pass results."""
    from lvae import estimators as E
    rng = np.random.default_rng(7)
    N, M, D = 700, 1000, 32
    mu = rng.standard_normal((N, D)).astype(np.float32)
    logvar = (0.3 * rng.standard_normal((N, D)) - 2.0).astype(np.float32)
    z = rng.standard_normal((M, D)).astype(np.float32)
    z[100:180] += 40.0                      # far in every dimension
    z[500:520, 3] -= 25.0                   # far in one dimension only
    z[900:, :] *= 1e3                       # absurdly far: densities far below fp32 range in linear space
    ref, refpm = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=True, reduce_mean=False)
    for path in ("auto", "direct"):
        lq, lqpm = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True,
                                         reduce_mean=False, path=PATHS[path])
        np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, err_msg=path)
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=1e-3, err_msg=path)
    # tiny and huge variances: b_jd outside the fast pass's range in both directions
    logvar2 = logvar.copy()
    logvar2[:, 0] = -60.0
    logvar2[:, 1] = 100.0
    _, refpm2 = onp.approximate_log_q_z(mu, logvar2, z[:64], method="all", prod_marginals=True,
                                        reduce_mean=False)
    _, lqpm2 = E.approximate_log_q_z(dev(mu), dev(logvar2), dev(z[:64]), method="all", prod_marginals=True,
                                     reduce_mean=False, path=PATHS["direct"])
    np.testing.assert_allclose(lqpm2.cpu().numpy(), refpm2, rtol=1e-4)


@pytest.mark.parametrize("N,M,D", [(2100, 2100, 32), (2500, 1800, 64), (4096, 1100, 20), (2200, 2000, 100),
                                    (2048, 2048, 128)])
def test_auto_path_prod_marginals_matrix_pipe_and_its_gate(N, M, D):
    """From 2^22 pairs on the AUTO path runs the prod-marginals polynomial on the bf16 matrix pipe
    (pair_pm_bx.hip, three limbs per value) behind the same device-side conditioning test as the joint density;
    ill-conditioned inputs keep the direct packed-fp32 sweep.  Both against the oracle; the gate's choice is
    read off bit-equality with the forced direct path."""
    from lvae import estimators as E
    from lvae import _native as N_
    rng = np.random.default_rng(N + D)
    mu = rng.standard_normal((N, D)).astype(np.float32)
    logvar = (0.3 * rng.standard_normal((N, D)) - 2.0).astype(np.float32)
    z = (mu[rng.integers(0, N, M)] + 0.4 * rng.standard_normal((M, D))).astype(np.float32)
    z[M // 10: M // 10 + 40] += 40.0                     # far in every dimension -> redo by the running-max sweep
    z[M // 2: M // 2 + 11, min(3, D - 1)] -= 25.0        # far in one dimension only
    ref, refpm = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=True, reduce_mean=False)
    lq, lqpm = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True, reduce_mean=False)
    lqd, lqpmd = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=True,
                                       reduce_mean=False, path=N_.PATH_DIRECT)
    np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=2e-3)
    assert rel(lqpm.double().mean(), refpm.mean()) < 1e-5
    assert float((lqpm - lqpmd).abs().max()) <= 2e-3
    assert not torch.equal(lqpm, lqpmd)                  # a different kernel produced it
    # peaked posteriors away from the origin: conditioning estimate above the limit -> the direct forms run
    mu2 = (3.0 * rng.standard_normal((N, D))).astype(np.float32)
    logvar2 = (0.2 * rng.standard_normal((N, D)) - 10.0).astype(np.float32)
    z2 = (mu2[rng.integers(0, N, M)] + np.exp(-5.0) * rng.standard_normal((M, D))).astype(np.float32)
    ref2, refpm2 = onp.approximate_log_q_z(mu2, logvar2, z2, method="all", prod_marginals=True, reduce_mean=False)
    lq2, lqpm2 = E.approximate_log_q_z(dev(mu2), dev(logvar2), dev(z2), method="all", prod_marginals=True,
                                       reduce_mean=False)
    lq2d, lqpm2d = E.approximate_log_q_z(dev(mu2), dev(logvar2), dev(z2), method="all", prod_marginals=True,
                                         reduce_mean=False, path=N_.PATH_DIRECT)
    scale = max(1.0, float(np.abs(refpm2).max()))
    np.testing.assert_allclose(lqpm2.cpu().numpy(), refpm2, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(lq2.cpu().numpy(), ref2, rtol=1e-4, atol=1e-4 * scale)
    assert torch.equal(lqpm2, lqpm2d) and torch.equal(lq2, lq2d)


def test_auto_path_prod_marginals_variance_input_and_excluded_diagonal():
    """The matrix-pipe prod-marginals sweep behind the two other switches of the entry point: posterior spread
    given as variance (mutual_information.py's convention) and the excluded diagonal pair (which only concerns
    the joint density).  Checked against the forced direct path of the same call."""
    from lvae import estimators as E
    from lvae import _native as N_
    g = torch.Generator(device="cuda").manual_seed(3)
    n, D = 2100, 32
    mu = torch.randn(n, D, device="cuda", generator=g)
    lv = 0.4 * torch.randn(n, D, device="cuda", generator=g) - 1.5
    z = mu[torch.randint(0, n, (2300,), device="cuda", generator=g)] + 0.5 * torch.randn(2300, D, device="cuda", generator=g)
    a = E.log_qz_raw(mu, lv, z, 1.0, True)
    b = E.log_qz_raw(mu, lv.exp(), z, 1.0, True, spread_kind=N_.SPREAD_VAR)
    c = E.log_qz_raw(mu, lv.exp(), z, 1.0, True, spread_kind=N_.SPREAD_VAR, path=N_.PATH_DIRECT)
    for k in (0, 1):
        scale = float(a[k].abs().max())
        assert float((a[k] - b[k]).abs().max()) <= 2e-6 * scale
        assert float((b[k] - c[k]).abs().max()) <= 2e-6 * scale
    d = E.log_qz_raw(mu, lv, z[:n], 1.0, True, exclude_diag=True)
    e = E.log_qz_raw(mu, lv, z[:n], 1.0, True, exclude_diag=True, path=N_.PATH_DIRECT)
    for k in (0, 1):
        assert float((d[k] - e[k]).abs().max()) <= 2e-6 * float(e[k].abs().max())
    assert not torch.equal(d[1], e[1])


def test_q1_raises():
    from lvae import estimators as E
    g = load_golden("logqz_generic")
    with pytest.raises(AttributeError):
        E.approximate_log_q_z(dev(g["mu"]), dev(g["logvar"]), dev(g["z"]), prod_marginals=False,
                              reduce_mean=True)


@pytest.mark.parametrize("N_,M_", [(700, 500), (2200, 2000)])
def test_hip_graph_capture_and_replay(N_, M_):
    """Every entry point is asynchronous on the caller's stream, takes a caller-provided workspace and picks
    its path on the device, so a whole estimator call can be captured in a HIP graph and replayed on new
    inputs (here through torch.cuda.CUDAGraph, which only provides the capture stream and the memory pool).
    At the larger size the captured call holds both prod-marginals sweeps (matrix pipe / packed fp32) and the
    replays take one or the other from the conditioning of the inputs."""
    from lvae import estimators as E
    rng = np.random.default_rng(11)
    D = 32
    static = [torch.zeros(N_, D, device="cuda"), torch.zeros(N_, D, device="cuda"), torch.zeros(M_, D, device="cuda")]
    E.log_qz_raw(static[0], static[1], static[2], 3.0, True)          # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lq, lqpm = E.log_qz_raw(static[0], static[1], static[2], 3.0, True)
    # the second input set takes the redo / fallback branches, the third (peaked posteriors away from the origin)
    # fails the conditioning test: direct kernels for both outputs
    for trial, (shift, peaked) in enumerate(((0.0, False), (30.0, False), (0.0, True))):
        mu = rng.standard_normal((N_, D)).astype(np.float32) * (3.0 if peaked else 1.0)
        lv = (0.5 * rng.standard_normal((N_, D)) - (10.0 if peaked else 1.0)).astype(np.float32)
        z = (rng.standard_normal((M_, D)) + shift).astype(np.float32)
        if peaked:
            z = (mu[rng.integers(0, N_, M_)] + np.exp(-5.0) * rng.standard_normal((M_, D))).astype(np.float32)
        for t, a in zip(static, (mu, lv, z)):
            t.copy_(torch.from_numpy(a))
        graph.replay()
        torch.cuda.synchronize()
        ref, refpm = onp.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
        ref, refpm = ref - (3.0 - np.log(N_)), refpm - D * (3.0 - np.log(N_))
        atol = 1e-3 if not peaked else 1e-4 * float(np.abs(refpm).max())
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=1e-4, atol=atol, err_msg=f"trial {trial}")
        np.testing.assert_allclose(lqpm.cpu().numpy(), refpm, rtol=1e-4, atol=atol, err_msg=f"trial {trial}")


def test_heterogeneous_posteriors_skip_fixed_reference():
    """A few extremely peaked posteriors among broad ones put the largest density normaliser far above the
    typical one: the fixed-reference pass would underflow for almost every sample, the device-side test
    switches it off and the running-maximum kernel does the whole job.  Also the in-between case where
    only some column groups are redone."""
    from lvae import estimators as E
    rng = np.random.default_rng(3)
    N_, D = 900, 32
    for n_peaked, lv_peaked in ((5, -14.0), (450, -9.0)):
        mu = rng.standard_normal((N_, D)).astype(np.float32)
        logvar = (0.3 * rng.standard_normal((N_, D)) - 0.5).astype(np.float32)
        logvar[:n_peaked] = lv_peaked
        mu[:n_peaked] = 1e-3 * rng.standard_normal((n_peaked, D))     # peaked AND near the origin: well conditioned
        z = (mu + rng.standard_normal((N_, D)) * np.exp(0.5 * logvar)).astype(np.float32)
        ref, _ = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False)
        lq, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                      reduce_mean=False, path=PATHS["mfma"])
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=2e-4, atol=2e-3, err_msg=str(n_peaked))
        lq, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                      reduce_mean=False, path=PATHS["auto"])
        np.testing.assert_allclose(lq.cpu().numpy(), ref, rtol=2e-4, atol=2e-3, err_msg=str(n_peaked))


def test_translation_and_permutation_invariance():
    """Size-independent properties at BASELINE's full size (configs[1]: N = M = 65,536, D = 32), where the
    oracle would not finish: log q(z) only depends on
    z - mu (a common translation of both -- here +50 in every dimension -- is removed by the centering shift
    before the fp32 contraction), and on the SET of posteriors (their order is irrelevant)."""
    from lvae import estimators as E
    g = torch.Generator(device="cuda").manual_seed(5)
    N_, M_, D = 65536, 65536, 32
    mu = torch.randn(N_, D, device="cuda", generator=g)
    lv = 0.5 * torch.randn(N_, D, device="cuda", generator=g) - 1.0
    z = mu[:M_] + torch.randn(M_, D, device="cuda", generator=g) * torch.exp(0.5 * lv[:M_])
    base, base_pm = E.approximate_log_q_z(mu, lv, z, method="all", prod_marginals=True, reduce_mean=False)
    moved, moved_pm = E.approximate_log_q_z(mu + 50.0, lv, z + 50.0, method="all", prod_marginals=True,
                                            reduce_mean=False)
    perm = torch.randperm(N_, device="cuda", generator=g)
    shuf, shuf_pm = E.approximate_log_q_z(mu[perm].contiguous(), lv[perm].contiguous(), z, method="all",
                                          prod_marginals=True, reduce_mean=False)
    for a, b, tol in ((moved, base, 2e-3), (moved_pm, base_pm, 2e-3), (shuf, base, 2e-4), (shuf_pm, base_pm, 2e-4)):
        assert float((a - b).abs().max()) <= tol, float((a - b).abs().max())
        assert abs(float(a.double().mean() - b.double().mean())) <= 1e-4 * abs(float(b.double().mean()))
    # and the sample under its own posterior bounds the mixture from below: log q(z_i) >= log q(z_i|x_i) - log N
    own = E.sample_log_likelihood(z, mu[:M_].contiguous(), lv[:M_].contiguous(), reduce_batch_dim=False)
    assert bool((base >= own - np.log(N_) - 1e-3).all())


def test_partial_logsumexp_blocks_merge_to_the_full_result():
    """lvae_log_qz_partial_f32 over disjoint blocks of posteriors + lvae_lse_combine_f32 (the pieces behind the
    overlapped multi-GPU exchange and the column-sharded variant) against the one-call result and the oracle.
    Covers blocks that take different branches: a well-behaved block, a block far from the samples (its
    fixed-reference pass underflows and is redone), and a block of peaked posteriors (direct path under AUTO)."""
    from lvae import estimators as E
    rng = np.random.default_rng(21)
    N_, M_, D = 1100, 700, 32
    mu = rng.standard_normal((N_, D)).astype(np.float32)
    logvar = (0.4 * rng.standard_normal((N_, D)) - 1.0).astype(np.float32)
    mu[400:650] += 30.0                              # far block
    logvar[650:700] = -12.0                          # peaked block
    z = (mu[rng.integers(0, 400, M_)] + 0.5 * rng.standard_normal((M_, D))).astype(np.float32)
    ref, _ = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False)
    full, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                    reduce_mean=False)
    np.testing.assert_allclose(full.cpu().numpy(), ref, rtol=1e-4, atol=1e-3)
    for bounds in ((0, 400, 650, 700, 1100), (0, 1, 1100), (0, 1100)):
        acc = None
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            part = E.log_qz_partial(dev(mu[lo:hi]), dev(logvar[lo:hi]), dev(z))
            acc = part if acc is None else E.lse_combine(acc, part)
        got = E.lse_combine(acc, None, np.log(N_))
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-3, err_msg=str(bounds))
        np.testing.assert_allclose(got.cpu().numpy(), full.cpu().numpy(), rtol=0, atol=2e-4, err_msg=str(bounds))
    # the merged pair is kept normalised (sum in [1, 2)) so that chains of merges stay in float range
    m, s = acc
    assert float(s.min()) >= 1.0 and float(s.max()) < 2.0
    # two-block form used by distributed.overlapped_stats
    own = E.log_qz_partial(dev(mu[:275]), dev(logvar[:275]), dev(z))
    rest = E.log_qz_partial(dev(mu[275:]), dev(logvar[275:]), dev(z))
    two = E.lse_combine(own, rest, np.log(N_))
    np.testing.assert_allclose(two.cpu().numpy(), ref, rtol=1e-4, atol=1e-3)
    # all-reduce form of the column-sharded variant (distributed.combine_partial_lse): max over the owners, every
    # owner's sums re-expressed against it on the device (lvae_lse_rebase_f32), summed, finished on the device
    from lvae.distributed import HipLseKernels as LK
    parts = [E.log_qz_partial(dev(mu[lo:hi]), dev(logvar[lo:hi]), dev(z)) for lo, hi in ((0, 400), (400, 700), (700, 1100))]
    gmax = torch.stack([p_[0] for p_ in parts]).max(0).values              # what all-reduce(MAX) leaves on every rank
    gsum = sum(LK.rebase(p_[0], p_[1], gmax) for p_ in parts)               # what all-reduce(SUM) leaves
    three = LK.finish(gmax, gsum, np.log(N_))
    np.testing.assert_allclose(three.cpu().numpy(), ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n_base,reps,D,M_", [(4096, 256, 64, 384),        # N = 1,048,576, D = 64: C3's posterior count
                                              (4096, 4097, 32, 256)])      # N = 16,781,312 > 2^24: no fixed reference
def test_large_posterior_counts(n_base, reps, D, M_):
    """Scale test of the posterior axis (64-bit indexing, split planning, and -- above 2^24 posteriors -- the
    running-maximum-only path).  The aggregate posterior of a k-fold replicated set equals that of the base
    set (log q is a mean over posteriors), so the big call is checked against the small one and the small one
    against the oracle."""
    from lvae import estimators as E
    rng = np.random.default_rng(n_base + reps)
    mu = rng.standard_normal((n_base, D)).astype(np.float32)
    logvar = (0.5 * rng.standard_normal((n_base, D)) - 1.0).astype(np.float32)
    z = (mu[:M_] + rng.standard_normal((M_, D)) * np.exp(0.5 * logvar[:M_])).astype(np.float32)
    ref, _ = onp.approximate_log_q_z(mu, logvar, z, method="all", prod_marginals=False, reduce_mean=False)
    small, _ = E.approximate_log_q_z(dev(mu), dev(logvar), dev(z), method="all", prod_marginals=False,
                                     reduce_mean=False)
    np.testing.assert_allclose(small.cpu().numpy(), ref, rtol=1e-4, atol=1e-3)
    mu_big, lv_big = dev(mu).repeat(reps, 1), dev(logvar).repeat(reps, 1)
    assert mu_big.shape[0] == n_base * reps
    big, _ = E.approximate_log_q_z(mu_big, lv_big, dev(z), method="all", prod_marginals=False, reduce_mean=False)
    np.testing.assert_allclose(big.cpu().numpy(), small.cpu().numpy(), rtol=0, atol=5e-4)
    del mu_big, lv_big
    torch.cuda.empty_cache()
