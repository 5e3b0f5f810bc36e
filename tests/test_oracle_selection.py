"""CPU: the oracle restatements behind the (f)-row additions -- Categorical MI bounds against the golden vectors
produced by the reference, the KS statistic against scipy.stats.ks_2samp (the third-party routine the reference
calls at utils_train.py:496-497; scipy 1.15.3 here), and the Pareto filter (host logic of the product, and its
brute-force oracle; the reference's utils_train does not import here -- transformers 4.2.2 internals -- so that
one is "parity unpinned" against the reference and pinned by the definition instead)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle_np as onp


@pytest.mark.parametrize("name", ["k64_v500", "k128_v600_peaked"])
def test_oracle_generative_mi_golden(name):
    g = load_golden("mi_generative_" + name)
    P = onp.categorical_log_prob_pairs(g["batch_probs"].reshape(-1, g["batch_probs"].shape[-1]),
                                       g["sampled_predictions"])
    np.testing.assert_allclose(P, g["log_prob_pairs"], rtol=2e-6, atol=2e-6)
    lower, upper = onp.calc_upper_and_lower_bound_generative_mi(g["batch_probs"], g["sampled_predictions"])
    assert abs(lower - float(g["lower"])) < 1e-4 * max(1.0, abs(float(g["lower"])))
    assert abs(upper - float(g["upper"])) < 1e-4 * max(1.0, abs(float(g["upper"])))


@pytest.mark.parametrize("n1,n2,shift,ties", [(50, 50, 0.0, False), (3370, 3370, 0.3, False), (1000, 777, 1.0, True),
                                              (1, 5, 0.0, False), (4096, 5000, 0.05, True)])
def test_oracle_ks_statistic_matches_scipy(n1, n2, shift, ties):
    from scipy import stats
    rng = np.random.default_rng(n1 + n2)
    a = rng.standard_normal(n1).astype(np.float32) * 20 - 100
    b = (rng.standard_normal(n2).astype(np.float32) * 20 - 100 + shift * 20).astype(np.float32)
    if ties:
        a, b = np.round(a), np.round(b)
    d = onp.ks_2samp_statistic(a, b)
    assert d == stats.ks_2samp(a, b, method="asymp").statistic     # the raw max |F_a - F_b|
    assert abs(d - stats.ks_2samp(a, b).statistic) < 1e-15         # 'exact' mode re-rounds it to a multiple of 1/lcm


def test_pareto_filter_host_logic():
    from lvae.pareto import is_pareto_efficient_simple, prepare_pareto_dict
    rng = np.random.default_rng(0)
    for n, d in ((1, 3), (12, 2), (40, 3), (25, 5)):
        pts = np.round(rng.standard_normal((n, d)), 1)            # rounding creates ties and duplicates
        pts[n // 2] = pts[0]
        got = is_pareto_efficient_simple(pts)
        np.testing.assert_array_equal(got, onp.is_pareto_efficient(pts))
        # definition: every kept point is undominated; every dropped point is weakly dominated by a kept one
        for i in range(n):
            dom = [j for j in range(n) if j != i and np.all(pts[j] >= pts[i]) and np.any(pts[j] > pts[i])]
            if got[i]:
                assert not dom
            else:
                assert any(got[j] and np.all(pts[j] >= pts[i]) for j in range(n) if j != i)
    assert list(is_pareto_efficient_simple(np.array([[1., 2.], [2., 1.], [.5, .5], [2., 1.]]))) == [True, True, False, False]

    class Cfg:
        pareto_check_use_rate = True
        pareto_check_use_tts_mmd = True
        pareto_check_use_elbo = False
    assert list(prepare_pareto_dict(Cfg()).keys()) == ["kl_analytical", "-tts_mmd"]


def test_pareto_filter_pinned_to_the_reference():
    """tests/golden/pareto.npz: gain matrices (ties, duplicates, a dominated point ahead of its dominator, one column,
    one point) and the masks the REFERENCE's is_pareto_efficient_simple returned for them (utils_train.py:416-431,
    run from its own FunctionDef by tests/golden/make_golden.py) -- both the product's vectorised filter and the
    oracle's brute-force restatement must reproduce every mask."""
    from conftest import load_golden
    from lvae.pareto import is_pareto_efficient_simple
    g = load_golden("pareto")
    n_cases = int(g["n_cases"])
    assert n_cases >= 8
    for k in range(n_cases):
        gains, want = g[f"gains_{k}"], g[f"efficient_{k}"]
        assert want.dtype == bool and want.shape == (gains.shape[0],)
        np.testing.assert_array_equal(is_pareto_efficient_simple(gains), want, err_msg=f"case {k}")
        np.testing.assert_array_equal(onp.is_pareto_efficient(gains), want, err_msg=f"oracle, case {k}")
