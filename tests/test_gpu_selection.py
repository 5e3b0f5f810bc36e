"""GPU parity of the (f)-row additions: Categorical (generative) MI bounds against the reference's golden vectors
and the oracle; the KS distance between IW-LL vectors against scipy (bit-exact) and the oracle; the Pareto
checkpoint bookkeeping end to end."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["k64_v500", "k128_v600_peaked"])
def test_generative_mi_golden(name):
    from lvae import mutual_information as MI
    g = load_golden("mi_generative_" + name)
    probs = torch.from_numpy(g["batch_probs"]).cuda()
    draw = torch.from_numpy(g["sampled_predictions"]).cuda()
    lower, upper = MI.calc_upper_and_lower_bound_generative_mi(probs, sampled_predictions=draw)
    assert abs(lower - float(g["lower"])) < 1e-4 * max(1.0, abs(float(g["lower"])))
    assert abs(upper - float(g["upper"])) < 1e-4 * max(1.0, abs(float(g["upper"])))
    # the reference's optional list of torch distributions
    dists = [torch.distributions.Categorical(probs=probs[i, 0]) for i in range(probs.shape[0])]
    l2, u2 = MI.calc_upper_and_lower_bound_generative_mi(probs, cat_dists=dists, sampled_predictions=draw)
    assert abs(l2 - lower) < 1e-5 and abs(u2 - upper) < 1e-5


def test_generative_mi_at_scale_and_errors():
    from lvae import _native as N
    from lvae import mutual_information as MI
    rng = np.random.default_rng(5)
    K, V = 1500, 50265                                        # roberta-base vocabulary
    probs = rng.random((K, V), dtype=np.float32) ** 8
    probs[np.arange(K), rng.integers(0, V, K)] += 5.0
    draw = rng.integers(0, V, K)
    lo_ref, up_ref = onp.calc_upper_and_lower_bound_generative_mi(probs, draw)
    lower, upper = MI.calc_upper_and_lower_bound_generative_mi(torch.from_numpy(probs).cuda().unsqueeze(1),
                                                               sampled_predictions=torch.from_numpy(draw))
    assert abs(lower - lo_ref) < 1e-4 * max(1.0, abs(lo_ref)) and abs(upper - up_ref) < 1e-4 * max(1.0, abs(up_ref))
    # the draw inside (torch.multinomial on the device): bounds are ordered and finite
    lower, upper = MI.calc_upper_and_lower_bound_generative_mi(torch.from_numpy(probs).cuda().unsqueeze(1))
    assert np.isfinite(lower) and np.isfinite(upper) and lower <= upper + 1e-6
    with pytest.raises(RuntimeError):                         # seq_len > 1 does not broadcast in the reference
        MI.calc_upper_and_lower_bound_generative_mi(torch.rand(4, 3, 10).cuda())
    bad = torch.tensor([0, 1, 2, 10])                         # class index out of range: NaN, no wild read
    lower, upper = MI.calc_upper_and_lower_bound_generative_mi(torch.rand(4, 1, 10).cuda(), sampled_predictions=bad)
    assert np.isnan(lower) and np.isnan(upper)
    with pytest.raises(N.LvaeError):
        MI.calc_upper_and_lower_bound_generative_mi(torch.rand(1, 1, 10).cuda())


@pytest.mark.parametrize("n1,n2,shift,ties", [(50, 50, 0.0, False), (3370, 3370, 0.3, False), (1000, 777, 1.0, True),
                                              (1, 5, 0.0, False), (4096, 4097, 0.05, True), (16384, 16384, 0.1, False),
                                              (300000, 70001, 0.02, False)])
def test_ks_statistic_bit_exact(n1, n2, shift, ties):
    from scipy import stats
    from lvae.pareto import ks_2samp
    rng = np.random.default_rng(n1 + n2)
    a = rng.standard_normal(n1).astype(np.float32) * 20 - 100
    b = (rng.standard_normal(n2).astype(np.float32) * 20 - 100 + shift * 20).astype(np.float32)
    if ties:
        a, b = np.round(a), np.round(b)
    d, p = ks_2samp(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert d == onp.ks_2samp_statistic(a, b)
    assert d == stats.ks_2samp(a, b, method="asymp").statistic     # the raw max |F_a - F_b|, bit for bit
    ref = stats.ks_2samp(a, b)                                      # 'exact' mode re-rounds to a multiple of 1/lcm
    assert abs(d - ref.statistic) < 1e-15
    if min(n1, n2) >= 1000:                                   # the limiting form is only close for large samples
        assert abs(p - ref.pvalue) < 0.02
    assert ks_2samp(a, a)[0] == 0.0                           # host arrays are accepted too (moved to the GPU)
    if n1 >= 50:                                              # a NaN propagates, like scipy's default nan_policy
        bad = a.copy()
        bad[n1 // 3] = np.nan
        assert all(np.isnan(v) for v in ks_2samp(bad, b)) and all(np.isnan(v) for v in ks_2samp(b, bad))
        assert ks_2samp(a, b)[0] == d                         # (the flag does not outlive the call)


def test_determine_pareto_checkpoint():
    from lvae.pareto import determine_pareto_checkpoint, ks_2samp
    rng = np.random.default_rng(1)
    track = {"kl_analytical": [], "iw_ll_p_w": [], "-distortion": [], "-tts_mmd": []}
    epochs = []
    for epoch in range(6):
        stats_e = {"kl_analytical": rng.random(10) + 0.1 * epoch, "iw_ll_p_w": -100 + rng.random(10),
                   "reconstruction_loss": 90 - epoch + rng.random(10), "tts_mmd_loss": rng.random(10) * 0.01}
        epochs.append(stats_e)
        records = []
        track, eff = determine_pareto_checkpoint(stats_e, track, epoch, logging=records.append)
        # the record of utils_train.py:510-514 ("pareto epoch" + this epoch's value of every tracked metric)
        assert records == [dict({"pareto epoch": epoch}, **{f"pareto {m}": track[m][-1] for m in track})]
    assert determine_pareto_checkpoint.__defaults__[0] is True        # the reference's default (:464)
    pts = np.array([[np.mean(e["kl_analytical"]), np.mean(e["iw_ll_p_w"]), -np.mean(e["reconstruction_loss"]),
                     -np.mean(e["tts_mmd_loss"])] for e in epochs])
    assert eff == np.where(onp.is_pareto_efficient(pts))[0].tolist()
    assert len(track["kl_analytical"]) == 6
    # "-D_ks" replaces its list by a scalar (utils_train.py:499) and the point matrix cannot be built: same here
    with pytest.raises(IndexError):
        determine_pareto_checkpoint({"iw_ll_p_w": rng.random(100), "iw_ll_x_gen_p_w": rng.random(100)},
                                    {"-D_ks": []}, 0)
    assert 0.0 <= ks_2samp(rng.random(100), rng.random(100) + 0.2)[0] <= 1.0
